def _pad_with_singletons(obj, num_singletons_before=0, num_singletons_after=0):
    shape = [1] * num_singletons_before + list(obj.shape) + [1] * num_singletons_after
    return obj.view(*shape)

"""Flow *spec generators* with the reference's call signatures and list-of-(name, init_dict) output format
(code/dsp/flows.py: SAL :115-136, StepTanhL :239-277).  Only the generators main.py can reach are provided."""
import numpy
import torch

from .utils import inv_softplus


def _common(options):
    return (options.get("set_res", False), options.get("add_f0", False), options.get("init_random", False),
            options.get("constraint", None))


def _input_dependent(options):
    dep = bool(options.get("input_dependent", False))
    if dep:
        assert "input_dim" in options, "You set to use input_dependent flows but the input dimension is not provided."
    cfg = {k: options[k] for k in ("batch_norm", "dropout", "hidden_dim", "hidden_activation", "num_hidden_layers",
                                   "inference") if k in options}
    return dep, options.get("input_dim", -1), cfg


def SAL(num_blocks, **kwargs):
    """[sinh_arcsinh, affine] x num_blocks; default init a=0,b=1 / a=1,b=0 is the identity map."""
    set_res, addf0, init_random, _ = _common(kwargs)
    dep, input_dim, cfg = _input_dependent(kwargs)
    blocks = []
    for _ in range(num_blocks):
        if init_random:
            a_aff, b_aff = numpy.random.randn(2)
            a_sal, b_sal = numpy.random.randn(2)
        else:
            a_aff, b_aff, a_sal, b_sal = 1.0, 0.0, 0.0, 1.0
        blocks.append(("sinh_arcsinh", {"init_a": a_sal, "init_b": b_sal, "add_init_f0": addf0,
                                        "set_restrictions": set_res, "input_dependent": dep, "input_dim": input_dim,
                                        "input_dependent_config": cfg}))
        blocks.append(("affine", {"init_a": a_aff, "init_b": b_aff, "set_restrictions": set_res}))
    return blocks


def StepTanhL(num_blocks, num_steps, **kwargs):
    """[step_flow(num_steps x tanh), affine] x num_blocks; needs the identity initialiser (initializers.py)."""
    _, addf0, init_random, _ = _common(kwargs)
    if "set_res" in kwargs:
        assert kwargs["set_res"] is True, "In the step tanh flow set_res has to be True for num_steps > 1"
    dep, input_dim, cfg = _input_dependent(kwargs)
    blocks = []
    for _ in range(num_blocks):
        steps = []
        for _s in range(num_steps):
            e1, e2, e3, e4 = numpy.random.randn(4)
            if not init_random:
                e2 = inv_softplus(torch.abs(torch.tensor((e2 + 1.0) / float(num_steps)))).item()
                e4 = inv_softplus(torch.abs(torch.tensor((e4 + 1.0) / float(num_steps)))).item()
            steps.append(("tanh", {"init_a": e1, "init_b": e2, "init_c": e3, "init_d": e4, "add_init_f0": False,
                                   "set_restrictions": True, "input_dependent": dep, "input_dim": input_dim,
                                   "input_dependent_config": cfg}))
        a_aff, b_aff = numpy.random.randn(2) if init_random else (1.0, 0.0)
        blocks.append(("step_flow", {"flow_arr": steps, "add_init_f0": addf0}))
        blocks.append(("affine", {"init_a": a_aff, "init_b": b_aff, "set_restrictions": False}))
    return blocks

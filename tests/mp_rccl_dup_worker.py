"""Worker of tests/test_gpu_multirank.py::test_abi_bootstrap_with_two_ranks_on_one_gpu_errors_out (not collected).

Two gloo ranks share the box's one GPU and build the C ABI's RCCL communicator (engine.RcclComm) over the multi-rank code
path: agreement step, id broadcast, store rendezvous, ncclCommInitRank under the watchdog.  RCCL refuses two ranks on one
device, so the init FAILS -- what is tested is HOW: every rank gets an exception (RuntimeError / TimeoutError with
tgp_last_error()), nobody hangs, nothing re-execs.  Prints BOOTSTRAP_ENDED <kind> per rank."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tgp.pytorch_amd import engine
    t0 = time.time()
    kind = "built"
    try:
        with torch.cuda.device(0):
            comm = engine.RcclComm(world, rank, None, timeout_s=60)
        comm.close()
    except engine.RcclUnavailable as e:
        kind = "unavailable: %s" % e
    except TimeoutError as e:
        kind = "timeout: %s" % e
    except RuntimeError as e:
        kind = "error: %s" % e
    sys.stdout.write("BOOTSTRAP_ENDED rank %d after %.1f s: %s\n" % (rank, time.time() - t0, kind[:300]))
    sys.stdout.flush()
    torch.distributed.destroy_process_group()
    os._exit(0)      # (a watchdog thread may still sit in ncclCommInitRank on a timeout: leave without joining it)


if __name__ == "__main__":
    main()

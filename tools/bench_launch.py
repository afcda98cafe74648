"""bench.py without a launcher: `python bench.py --gpus N` starts its own ranks (launch_ranks), `--launch-check` rehearses the
rendezvous and the ranks' call plans on CPU tensors (launch_check).  Imported by bench.py before any GPU call."""
import json
import os
import sys

import torch

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")

def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torchrun, rendezvous on
    127.0.0.1), let rank 0's JSON line through on stdout and exit with the child's code.  Runs before this process has made
    any GPU call and makes none (torch.cuda.device_count() does not initialise the device on this image): the parent never
    replaces itself with another program, it waits.  Fewer visible GPUs than ranks is an error -- never a silent one-GPU run."""
    import socket
    import subprocess
    if not a.launch_check:
        need, have = (1 if a.single_device else a.gpus), torch.cuda.device_count()
        if have < need:
            print(f"bench.py --gpus {a.gpus}: needs {need} visible GPU(s), found {have} (one rank per GPU; --single-device "
                  f"--dist-backend gloo rehearses the multi-rank path on one)", file=sys.stderr)
            sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.run(cmd, env=env).returncode           # stdout / stderr inherited: rank 0 prints the line
    sys.exit(rc if rc >= 0 else 1)


def launch_check(a, rank, world, WORKLOADS, plan_calls):
    """--launch-check: the ranks' rendezvous, view shares and call plans without any GPU work (gloo, CPU tensors)."""
    import torch.distributed as dist
    from view_sharding import views_of_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_vox, n_views, W, H, C = WORKLOADS[a.workload if a.workload in WORKLOADS else "R2"]
    n_views = a.views or n_views
    mine = views_of_rank(n_views, rank, world)
    plan = plan_calls(len(mine), H, W, C, 4 if a.dtype == "f32" else 2, a.chunk, a.call_gb,
                      a.min_calls if a.min_calls is not None else (1 if world > 1 else 2), a.pool) if mine else (0, 0, 0)
    t = torch.zeros(world, 3, dtype=torch.int64)
    t[rank] = torch.tensor([len(mine), plan[0], plan[1]])
    dist.all_reduce(t)
    if rank == 0:
        assert int(t[:, 0].sum()) == n_views
        print(json.dumps({"metric": "Mvoxel-views/sec", "value": None, "unit": "Mvoxel-views/s", "n_gpus": dist.get_world_size(),
                          "launch_check": True, "gpus_requested": a.gpus, "views_per_rank": t[:, 0].tolist(),
                          "views_per_call": t[:, 1].tolist(), "calls_per_rank": t[:, 2].tolist(),
                          "config": {"workload": a.workload}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()

"""The N>1 path on CPU: two processes over the gloo backend run the same plumbing
bench.py uses on RCCL -- one broadcast of the shared parameter block from rank 0,
static sharding of independent frame pairs, max-over-ranks timing and the
whole-job aggregate.  (No collective sits on the data path.)"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from videomorphing_amd import capi, dist as vdist, morph, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    blk = capi.ParamBlock()
    cons = None
    if rank == 0:                      # only rank 0 knows the parameters
        P = morph.Parameters(); P.bcond = capi.BCOND_BORDER
        blk.kp = morph.KernParameters(P)
        blk.max_iter, blk.max_iter_drop_factor, blk.start_res, blk.math_mode = 500.0, 1.0, 32, 1
        cons = synth.make_constraints(1920, 1080, %d)
    raw = vdist.broadcast_block(vdist.pack_block(blk, cons), torch.device("cpu"))
    blk, cons = vdist.unpack_block(raw)
    mine = vdist.shard_pairs(7, world, rank)
    units = float(sum(1000 + k for k in mine))      # stand-in for pixel*iters of my pairs
    elapsed = 1.0 + rank                              # rank 1 is the slow one
    tmax, total = vdist.reduce_report(elapsed, units, torch.device("cpu"))
    dist.barrier()
    open(os.path.join(%r, "rank%%d.json" %% rank), "w").write(json.dumps({"rank": rank, "bcond": blk.kp.bcond, "max_iter": blk.max_iter, "math": blk.math_mode,
                      "ncons": int(len(cons)), "cons0": cons[0].tolist(), "mine": mine,
                      "tmax": tmax, "total": total}))
    dist.destroy_process_group()
""")


import pytest


@pytest.mark.parametrize("ncons", [8, 300])      # 300: more than any fixed-size staging buffer of old
def test_two_rank_gloo_broadcast_shard_and_reduce(tmp_path, ncons):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ncons, str(tmp_path)))
    from videomorphing_amd.launch import free_port
    port = str(free_port())                 # an ephemeral port: no clash with other jobs on the host
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    recs = [json.load(open(str(tmp_path / ("rank%d.json" % r)))) for r in range(2)]
    assert [r["rank"] for r in recs] == [0, 1]
    from videomorphing_amd import synth
    want = synth.make_constraints(1920, 1080, ncons)[0].tolist()
    for r in recs:                                  # both ranks hold rank 0's block
        assert r["bcond"] == 2 and r["max_iter"] == 500.0 and r["math"] == 1
        assert r["ncons"] == ncons and r["cons0"] == want
        assert r["tmax"] == 2.0                     # MAX over ranks
        assert r["total"] == sum(1000 + k for k in range(7))   # SUM over ranks: whole job
    assert recs[0]["mine"] + recs[1]["mine"] == list(range(7))


SELF = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    from videomorphing_amd import launch
    n = int(sys.argv[sys.argv.index("--gpus") + 1])
    if launch.needs_launch(n):
        # the parent: no torch import, no GPU call -- children only
        assert "torch" not in sys.modules
        sys.exit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], n))
    import torch.distributed as dist
    dist.init_process_group("gloo")
    assert dist.get_world_size() == n
    t = __import__("torch").tensor([dist.get_rank() + 1.0])
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print("RESULT world=%%d sum=%%g" %% (n, t.item()), flush=True)
    dist.destroy_process_group()
    sys.exit(7 if "--fail" in sys.argv else 0)
""")


def test_self_launch_spawns_n_ranks_and_relays_output_and_exit_code(tmp_path):
    """`python script.py --gpus 2` without torch.distributed.run in front (the form the
    driver uses for bench.py): the parent must start 2 ranks as children, pass rank 0's line
    through and exit with their code."""
    script = tmp_path / "selfl.py"
    script.write_text(SELF % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT world=2 sum=3" in out.stdout
    out = subprocess.run([sys.executable, str(script), "--gpus", "2", "--fail"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0

"""N > 1 launch logic on CPU (gloo, world_size 2): stream sharding, the weight-blob broadcast and the
max-over-ranks timing reduction that bench.py uses.  No compute: the HIP path needs a GPU."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, hashlib
    sys.path.insert(0, os.path.join(%r, "fast-artistic-videos_amd", "python"))
    import torch, torch.distributed as dist
    import fav_amd
    from fav_amd import t7, shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    path = os.path.join(%r, "tests", "golden", "tiny_model.t7")
    blob = shard.broadcast_blob(fav_amd.pack_checkpoint(path) if rank == 0 else None, torch.device("cpu"))
    digest = hashlib.sha256(blob).hexdigest()
    want = hashlib.sha256(fav_amd.pack_checkpoint(path)).hexdigest()
    assert digest == want, "rank %%d received a different weight blob" %% rank
    mine = shard.streams_for_rank(5, rank, world)
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    assert sorted(sum(allv, [])) == list(range(5)) and all(s %% world == r for r, ss in enumerate(allv) for s in ss)
    t = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == float(world)
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % (ROOT, ROOT)


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2

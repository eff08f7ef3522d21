"""Host logic of the node-sharded run (gpt-st_amd/shard.py) on CPU: parameter sharding round trip, and the label all-gather
of the adaptive-mask phase over a real 2-rank gloo group (rank-major gather -> (B,T,N) node order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import free_port  # noqa: E402
import torch.multiprocessing as mp

from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_state_dict_round_trip_and_classification():
    from gptst_amd.shard import is_node_local, is_replicated_compute, shard_state_dict, unshard_state_dicts
    args = make_args("PEMS08", num_nodes=24, embed_dim=4, HS=4, HT=4, scaler_zeros=synth.scaler_zeros())
    sd = O.init_state_dict(args, 3)
    local = [k for k in sd if is_node_local(k)]
    # 2 STHCNs x (node_embeddings, node_embeddings_spg, cap1.adj, cap2.adj) + encoder.neb4mask
    assert len(local) == 9 and all(24 in sd[k].shape for k in local), local
    assert all(24 not in sd[k].shape for k in sd if not is_node_local(k) and sd[k].dim() > 0)
    repl = [k for k in sd if is_replicated_compute(k)]
    assert len(repl) == 2 * (2 + 10), repl                       # per STHCN: cap1/cap2 t_adj + the 5 Linear layers of time_feature2
    parts = [shard_state_dict(sd, r * 8, (r + 1) * 8) for r in range(3)]
    for k in local:
        assert 8 in parts[1][k].shape and 24 not in parts[1][k].shape
    back = unshard_state_dicts(parts)
    assert all(torch.equal(back[k], sd[k]) for k in sd)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from gptst_amd.shard import DistNodeGroup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T, Nl = 2, 3, 4
    N = Nl * world
    glob = torch.arange(B * T * N, dtype=torch.int32).view(B, T, N)
    mine = glob[:, :, rank * Nl:(rank + 1) * Nl].contiguous()
    grp = DistNodeGroup(rank, world)
    lab = grp.all_gather(mine)
    label_g = lab.permute(1, 2, 0, 3).contiguous().view(B, T, N)          # what ShardedPretrainStep._mask does
    t = torch.full((5,), float(rank + 1))
    grp.all_reduce_(t)
    q.put((rank, label_g.tolist(), t.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_label_gather_restores_node_order_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.arange(2 * 3 * 8, dtype=torch.int32).view(2, 3, 8).tolist()
    for rank, lab, t in got:
        assert lab == want and t == [3.0] * 5

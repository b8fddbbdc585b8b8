"""Worker of tests/test_model_gpu.py::test_two_ranks_on_the_hip_path_match_single_process_sum_of_shards: one rank of a
`python -m torch.distributed.run` job whose ranks SHARE cuda:0 (gloo backend: RCCL refuses two ranks on one device; the one-GPU
box has no second device).  Everything else is the product path: HIP kernels through the C ABI, direct gradient writes into the
flat buffer, GradSync's per-source release counting, descending bucket order, MADGRAD on the device."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def global_batch(V):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 80, 256, generator=g)
    ln = torch.tensor([256, 200, 256, 232])
    tg = torch.randint(0, V, (4, 8), generator=g)
    tl = torch.tensor([8, 6, 7, 8])
    return x, ln, tg, tl


def main():
    out = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    from common_model import build_from_fixture
    from conftest import load_golden
    from lcasr_amd.parallel import broadcast_module_state
    from lcasr_amd.train import Trainer
    fx = load_golden('tiny_ln_ragged')
    m, V = build_from_fixture(fx, 'cuda'), int(fx['cfg.vocab_size'])
    broadcast_module_state(m)
    tr = Trainer(m, lr=3e-3, clip_value=0.8, global_batch=4, bucket_bytes=32 << 10)
    assert len(tr.sync.buckets) > 3
    tr.sync.profile = True
    x, ln, tg, tl = global_batch(V)
    sl = slice(rank * 2, rank * 2 + 2)
    losses = [float(tr.step(x[sl].cuda(), ln[sl].cuda(), tg[sl].cuda(), tl[sl].cuda())) for _ in range(3)]
    expect = dict(tr.sync._expect)
    assert all(expect[i][0] == 'd' for i in expect), expect               # every parameter is written directly by a backward kernel
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, data=tr.opt.flat[0].data.cpu(), wait_ms=tr.sync.exposed_wait_ms(),
                    nbt=int(m.layers[0].conv.fn.batch_norm.num_batches_tracked)), out + f'.{rank}')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

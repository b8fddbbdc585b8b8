"""world_size-2 `gloo` test (CPU) of the data-parallel machinery: FlatParams + GradSync bucketed, backward-overlapped
all-reduce.  The model here is a plain torch MLP (the HIP path needs a GPU); what is under test is the N>1 logic:
flat-buffer views, bucket boundaries, hook-driven async all-reduce, SUM semantics with a global loss normaliser."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(16, 33), torch.nn.Tanh(), torch.nn.Linear(33, 9), torch.nn.Tanh(), torch.nn.Linear(9, 5))


def _worker(rank, world, port, bucket_bytes, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lcasr_amd.optim import FlatParams
    from lcasr_amd.parallel import GradSync, broadcast_module_state
    m = _model()
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters(): p.add_(1.0)                 # diverge, then broadcast must repair it
    broadcast_module_state(m)
    fp = FlatParams(m.parameters())
    sync = GradSync(fp.params, fp.grad, fp.offsets, bucket_bytes=bucket_bytes)
    g = torch.Generator().manual_seed(100)
    X = torch.randn(8, 16, generator=g); Y = torch.randn(8, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]   # shard the global batch of 8
    for _ in range(2):                                           # two steps: hooks must re-arm
        fp.zero_grad()
        loss = ((m(xs) - ys) ** 2).sum() / 8                     # GLOBAL normaliser, SUM all-reduce
        loss.backward()
        sync.finish()
    if rank == 0:
        torch.save(dict(grad=fp.grad.clone(), n_buckets=len(sync.buckets), data=fp.data.clone()), out)
    dist.destroy_process_group()


class _DirectLinear(torch.autograd.Function):
    """A linear layer whose backward writes dW straight into weight.grad and announces it (what the HIP backward kernels
    do under functional.set_direct_grad): autograd never sees that gradient."""
    hook = None

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x); ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        ctx.w.grad.add_(dy.t() @ x)
        _DirectLinear.hook(ctx.w)
        return dy @ ctx.w, None


def _direct_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lcasr_amd.optim import FlatParams
    from lcasr_amd.parallel import GradSync
    torch.manual_seed(3)
    shared = torch.nn.Parameter(torch.randn(16, 16) * 0.3)       # used TWICE per forward: two direct writes per backward
    tail = torch.nn.Linear(16, 5)
    fp = FlatParams([shared] + list(tail.parameters()))
    sync = GradSync(fp.params, fp.grad, fp.offsets, bucket_bytes=64)   # tiny buckets: `shared` has a bucket of its own
    _DirectLinear.hook = sync.on_grad_ready
    g = torch.Generator().manual_seed(5)
    X = torch.randn(8, 16, generator=g); Y = torch.randn(8, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    for _ in range(3):                                           # step 1 learns the write counts, steps 2-3 overlap
        fp.zero_grad()
        h = torch.tanh(_DirectLinear.apply(torch.tanh(_DirectLinear.apply(xs, shared)), shared))
        (((tail(h) - ys) ** 2).sum() / 8).backward()
        sync.finish()
    if rank == 0:
        torch.save(dict(grad=fp.grad.clone()), out)
    dist.destroy_process_group()


def test_gradsync_direct_writes_shared_parameter(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_direct_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    torch.manual_seed(3)
    shared = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
    tail = torch.nn.Linear(16, 5)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(8, 16, generator=g); Y = torch.randn(8, 5, generator=g)
    h = torch.tanh(torch.tanh(X @ shared.t()) @ shared.t())
    (((tail(h) - Y) ** 2).sum() / 8).backward()
    ref = torch.cat([shared.grad.reshape(-1), tail.weight.grad.reshape(-1), tail.bias.grad.reshape(-1)])
    got = res['grad']
    assert torch.allclose(got[:256], ref[:256], atol=1e-5) and torch.allclose(got[256:256 + 80], ref[256:336], atol=1e-5)


@pytest.mark.parametrize('bucket_bytes', [256, 1 << 20])
def test_gradsync_matches_single_process(tmp_path, bucket_bytes):
    port = 29500 + (os.getpid() % 2000) + (1 if bucket_bytes == 256 else 0)
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, port, bucket_bytes, out), nprocs=2, join=True)
    res = torch.load(out)
    sys.path.insert(0, ROOT)
    from lcasr_amd.optim import FlatParams
    m = _model()
    fp = FlatParams(m.parameters())
    g = torch.Generator().manual_seed(100)
    X = torch.randn(8, 16, generator=g); Y = torch.randn(8, 5, generator=g)
    (((m(X) - Y) ** 2).sum() / 8).backward()
    assert torch.equal(res['data'], fp.data), 'broadcast did not equalise the replicas'
    assert torch.allclose(res['grad'], fp.grad, atol=1e-6), float((res['grad'] - fp.grad).abs().max())
    assert res['n_buckets'] == (1 if bucket_bytes > 4096 else res['n_buckets']) and res['n_buckets'] >= 1
    if bucket_bytes == 256:
        assert res['n_buckets'] > 2


def test_flat_params_views_and_zero_grad():
    sys.path.insert(0, ROOT)
    from lcasr_amd.optim import FlatParams
    m = _model()
    before = [p.detach().clone() for p in m.parameters()]
    fp = FlatParams(m.parameters())
    for p, b, o in zip(m.parameters(), before, fp.offsets):
        assert torch.equal(p, b) and p.data_ptr() == fp.data.data_ptr() + 4 * o and o % 4 == 0
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o
    m(torch.randn(3, 16)).sum().backward()
    assert float(fp.grad.abs().sum()) > 0
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0 and all(float(p.grad.abs().sum()) == 0 for p in m.parameters())

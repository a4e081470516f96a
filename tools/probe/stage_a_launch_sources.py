"""Which Python line launches which device kernel in a fresh stage-A iteration (torch.profiler with stacks): the torch-origin launches
(copies, fills, cats, index ops) with the innermost recon_amd / tools frame that issued them."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
import stage_a_iter_bench as sb

# build the same objects as the bench through its main() internals: run it once with a tiny iteration count to warm everything up
sb.main(["--iters", "2", "--loss-rows", "recon"])
# re-create the pieces (main() keeps them local): a second small instance
import numpy as np
from recon_amd.models import SpKBGATModified
from recon_amd.sampler import KGNeighbourSampler
from recon_amd.graph import trust
from recon_amd.losses import batch_gat_loss
dv = torch.device("cuda:0")
N, nrel = 14541, 237
adj_idx, adj_val = sb.synthetic_kg(N=N, nrel=nrel)
sampler = KGNeighbourSampler(adj_idx.to(dv), adj_val.to(dv), N)
torch.manual_seed(0)
model = SpKBGATModified(torch.randn(N, 50), torch.randn(nrel, 50), [100, 200], [100, 200], 0.3, 0.2, [2, 2]).to(dv).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-3)
loss_fn = torch.nn.MarginRankingLoss(margin=1.0)
src_all = torch.unique(adj_idx[1])


def it():
    ents = trust(src_all[torch.randperm(src_all.numel())[:128]].to(dv), bound=N)
    (edge, et), (srcs, _) = sampler.batch_adj_data(ents)
    quads = sampler.batch_nhop_neighbors(srcs)
    pos = torch.stack((edge[1], et, edge[0]), dim=1)
    neg = pos.repeat(4, 1)
    half = neg.shape[0] // 2
    neg[:half, 0] = torch.randint(0, N, (half,), device=dv)
    neg[half:, 2] = torch.randint(0, N, (neg.shape[0] - half,), device=dv)
    tri = trust(torch.cat((pos, neg), dim=0), bound=N, rel_bound=nrel)
    e, r, _ = model(None, ents, (edge, et), quads)
    opt.zero_grad()
    loss = batch_gat_loss(loss_fn, tri, e, r, valid_invalid_ratio_gat=2)
    loss.backward()
    opt.step()
    return loss.item()


for _ in range(3):
    it()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    it()
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::fill_", "aten::cat", "aten::index", "aten::index_put_", "aten::arange", "aten::add", "aten::add_", "aten::zero_", "aten::_to_copy",
        "aten::native_dropout", "aten::mul", "aten::stack", "aten::clone", "aten::zeros", "aten::index_select", "aten::repeat", "aten::_local_scalar_dense",
        "aten::constant_pad_nd", "aten::slice_backward", "aten::sum", "aten::neg", "aten::ones", "aten::ones_like", "aten::zeros_like")
cnt = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in want:
        frames = [f for f in (ev.stack or []) if "recon_amd/" in f or "stage_a_launch_sources" in f or "torch/optim" in f]
        where = frames[0].split("/root/repo/")[-1] if frames else ((ev.stack or ["?"])[0][-70:])
        cnt[(ev.key, where)] += ev.count
for (name, where), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-24s %s" % (c, name, where))

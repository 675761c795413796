"""Single-GPU C5 (2048^3 uint32, 100k seeds): size-independent checks at full scale."""
import numpy as np, torch, time
from tissue_analysis_amd import _capi, device as dev, synth
c = synth.CONFIGS["C5"]; dims = c["dims"]; dtype = np.dtype(c["dtype"])
ctx = dev.torch_context(0)
t0 = time.time(); vol, L = dev.synth_slab(ctx, dims, dtype, c["n_cells"], c["seed"]); torch.cuda.synchronize()
print("synth %.2fs" % (time.time() - t0), flush=True)
ctx.set_volume_device(vol.data_ptr(), 4, vol.shape, keep=vol)
for it in range(3):
    ctx.extract(_capi.F_ALL, L); ctx.synchronize(); t = ctx.timing()
    print("sweep %.3f ms total %.3f ms -> %.1f GB/s" % (t["ms_sweep"], t["ms_total"], t["bytes_read"] / t["ms_sweep"] / 1e6), ctx.debug_counters(), flush=True)
count, bbox, s1, s2 = ctx.labels(); lo, hi, f = ctx.adjacency()
nvox = int(np.prod(dims)); n = dims[0]
assert int(count.sum()) == nvox, (int(count.sum()), nvox)
for d in range(3):
    assert int(s1[:, d].sum()) == (nvox // n) * (n * (n - 1) // 2)
    assert int(s2[:, [0, 3, 5][d]].sum()) == (nvox // n) * ((n - 1) * n * (2 * n - 1) // 6)
present = count > 0
assert np.all(bbox[present, :3] >= 0) and np.all(bbox[present, 3:] <= n)
assert np.all(lo < hi) and np.all(np.diff((lo.astype(np.int64) << 32) | hi) > 0)
# cross-check the first 16 planes exactly against the C oracle
from oracle import onepass_c
sub = vol[:16].cpu().numpy().view(np.uint32)
want = onepass_c.extract(sub, max_label=L)
ctx.set_volume_device(vol.data_ptr(), 4, (16,) + tuple(dims[1:]), keep=vol)
ctx.extract(_capi.F_ALL, L); c2, b2, s12, s22 = ctx.labels(); l2, h2, f2 = ctx.adjacency()
for k, g in (("count", c2), ("bbox", b2), ("sum1", s12), ("sum2", s22), ("pair_lo", l2), ("pair_hi", h2), ("pair_faces", f2)):
    assert np.array_equal(g, want[k]), k
print("C5 ok: labels", int(present.sum()), "pairs", lo.size)

"""The adjacency exchange of the multi-GPU step without any process group: two contexts on one GPU hold the two Z-slabs
of a volume, their accumulators are reduced with plain torch ops, the exchange blocks are concatenated by hand.
ta_adjacency_pack (every pair travels) and ta_adjacency_pack_shared (only pairs a slab face can split) must both end in
the unsharded adjacency."""
import numpy as np
import pytest

from oracle import onepass_c
from tissue_analysis_amd import _capi, distributed as tad

from helpers import voronoi

pytestmark = pytest.mark.gpu


def _slab_job(vol, lo, hi, L):
    import torch
    halo = 1 if lo > 0 else 0
    part = np.ascontiguousarray(vol[lo - halo:hi])
    t = torch.from_numpy(part.view(np.int32).copy()).to("cuda:0")
    ctx = _capi.Context(0)
    sums = torch.zeros((L + 1, 10), dtype=torch.int64, device="cuda:0")
    boxes = torch.zeros((L + 1, 6), dtype=torch.int32, device="cuda:0")
    ctx.set_volume_device(t.data_ptr(), 4, part.shape, a0_origin=lo, has_low_halo=bool(halo), keep=t)
    ctx.bind_accumulators(sums.data_ptr(), boxes.data_ptr(), L, keep=(sums, boxes))
    return ctx, sums, boxes


@pytest.mark.parametrize("shared", [False, True], ids=["pack_all", "pack_shared"])
def test_two_slabs_exchange_blocks_by_hand(shared):
    import torch
    vol = voronoi((36, 40, 264), 70, 41, np.uint32)
    want = onepass_c.extract(vol)
    L = int(vol.max())
    cut = 17
    jobs = [_slab_job(vol, 0, cut, L), _slab_job(vol, cut, vol.shape[0], L)]
    try:
        for ctx, _, _ in jobs:
            ctx.extract(_capi.F_ALL, L)
        npairs = [ctx.adjacency_size() for ctx, _, _ in jobs]
        # per-label reduce, as the two all-reduces would do it
        sums = jobs[0][1] + jobs[1][1]
        boxes = torch.minimum(jobs[0][2], jobs[1][2])
        torch.cuda.synchronize()
        for _, s, b in jobs:
            s.copy_(sums); b.copy_(boxes)
        got = tad.from_device_layout(sums.cpu().numpy(), boxes.cpu().numpy())
        for k in ("count", "bbox", "sum1", "sum2"):
            assert np.array_equal(got[k], want[k]), k
        for ctx, _, _ in jobs:
            ctx.accumulators_reduced()
        cap = max(npairs) + 8
        words = _capi.exchange_words(cap)
        blocks = torch.empty((2 * words,), dtype=torch.int64, device="cuda:0")
        for r, (ctx, _, _) in enumerate(jobs):
            ptr = blocks[r * words:(r + 1) * words].data_ptr()
            (ctx.adjacency_pack_shared if shared else ctx.adjacency_pack)(ptr, cap)
        torch.cuda.synchronize()
        sent = [int(blocks[r * words].item()) for r in range(2)]
        if shared:
            assert all(s <= n for s, n in zip(sent, npairs)) and sum(sent) < sum(npairs), (sent, npairs)     # something stayed at home
        else:
            assert sent == npairs
        lists = []
        for ctx, _, _ in jobs:
            assert ctx.adjacency_scope() == _capi.ADJ_LOCAL
            ctx.adjacency_merge_blocks(blocks.data_ptr(), 2, cap)
            assert ctx.adjacency_scope() == (_capi.ADJ_PARTIAL if shared else _capi.ADJ_MERGED)
            if shared:                     # a partial list is handed out only when asked for by name
                with pytest.raises(_capi.TissueScanError):
                    ctx.adjacency()
            lists.append(ctx.adjacency(allow_partial=True))
        if not shared:                     # every rank holds the global list
            for lo, hi, faces in lists:
                assert np.array_equal(lo, want["pair_lo"]) and np.array_equal(hi, want["pair_hi"]) and np.array_equal(faces, want["pair_faces"])
        else:                              # private lists are disjoint; their union with the merged travelling pairs is global
            a0 = [0, cut]; a1 = [cut, vol.shape[0]]
            bx = boxes.cpu().numpy()
            merged = {}
            for r, (lo, hi, faces) in enumerate(lists):
                excl = tad.slab_exclusive(bx, a0[r], a1[r])
                for a, b, f in zip(lo.tolist(), hi.tolist(), faces):
                    private = bool(excl[a] or excl[b])
                    key = (a, b)
                    if private:
                        assert key not in merged, key
                        merged[key] = f
                    elif key in merged:
                        assert np.array_equal(merged[key], f), key          # the travelling part is the same on both ranks
                    else:
                        merged[key] = f
            keys = sorted(merged)
            assert keys == list(zip(want["pair_lo"].tolist(), want["pair_hi"].tolist()))
            assert np.array_equal(np.array([merged[k] for k in keys]), want["pair_faces"])
    finally:
        for ctx, _, _ in jobs:
            ctx.close()

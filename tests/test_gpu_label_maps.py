"""Label lookup-table sweeps on the GPU: fuse / remove labels in the image and the
per-label property image, against the oracle's restatement of the reference loops."""
import numpy as np
import pytest

from oracle import sia_oracle
from oracle.sia_oracle import OracleSIA
from tissue_analysis_amd import DICT, SpatialImage, SpatialImageAnalysis, _capi, synth

from api_compare import compare_api
from helpers import voronoi

pytestmark = pytest.mark.gpu
VS = synth.PARITY_VOXELSIZE


def pair(vol):
    a, b = vol.copy(), vol.copy()
    sia = SpatialImageAnalysis(SpatialImage(a, voxelsize=VS), ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(b, ignoredlabels=0, return_type=sia_oracle.DICT, background=1, voxelsize=VS)
    return sia, ref, a, b


def fresh_reference(ref, image):
    """What a new reference analysis of the mutated image answers, with the same ignored labels."""
    return OracleSIA(image, ignoredlabels=sorted(ref.ignoredlabels() - set([1])), return_type=sia_oracle.DICT,
                     background=1, voxelsize=VS)


@pytest.mark.parametrize("dtype,order", [(np.uint16, "C"), (np.uint32, "F")])
def test_fuse_labels(dtype, order):
    vol = np.asarray(voronoi((30, 34, 70), 40, 51, dtype), order=order)
    sia, ref, a, b = pair(vol)
    cells = [l for l in ref.labels() if l != 1]
    group = [cells[7], cells[2], cells[11], 60000 if dtype == np.uint32 else 50000]    # one absent label
    mine, theirs = list(group), list(group)
    assert sia.fuse_labels_in_image(mine, verbose=False) is None
    ref.fuse_labels_in_image(theirs)
    assert mine == theirs                                      # the minimum was removed from the caller's list
    assert np.array_equal(np.asarray(sia.image), np.asarray(ref.image))
    assert np.array_equal(a, b) and not np.array_equal(a, vol) # edited in place, like the reference
    compare_api(sia, fresh_reference(ref, b))                  # every later answer describes the new image


def test_remove_labels_and_margins():
    vol = voronoi((30, 34, 70), 40, 52, np.uint16)
    sia, ref, a, b = pair(vol)
    cells = [l for l in ref.labels() if l != 1]
    sia.remove_labels_from_image([cells[3], 1, cells[9]], erase_value=0, verbose=False)     # background is skipped
    ref.remove_labels_from_image([cells[3], 1, cells[9]], erase_value=0)
    assert np.array_equal(a, b) and (a == 0).any() and (a == 1).any()
    assert sia.ignoredlabels() == ref.ignoredlabels()
    compare_api(sia, fresh_reference(ref, b))
    sia.remove_labels_from_image(cells[5], erase_value=7, verbose=False)                    # scalar label, other erase value
    ref.remove_labels_from_image(cells[5], erase_value=7)
    assert np.array_equal(a, b) and sia.ignoredlabels() == ref.ignoredlabels()

    sia, ref, a, b = pair(vol)
    sia.remove_stack_margin_labels_from_image(verbose=False)
    ref.remove_stack_margin_labels_from_image()
    assert np.array_equal(a, b)
    compare_api(sia, fresh_reference(ref, b))
    with pytest.raises(ValueError):
        sia.remove_labels_from_image([cells[20]], erase_value=70000, verbose=False)          # does not fit uint16


def test_relabel_of_a_non_native_image_dtype():
    vol = voronoi((18, 20, 40), 12, 53, np.uint16).astype(np.int64)
    sia = SpatialImageAnalysis(vol, ignoredlabels=0, return_type=DICT, background=1)
    ref = OracleSIA(vol.copy(), ignoredlabels=0, return_type=sia_oracle.DICT, background=1)
    cells = [l for l in ref.labels() if l != 1]
    sia.remove_labels_from_image([cells[0], cells[4]], verbose=False)
    ref.remove_labels_from_image([cells[0], cells[4]])
    assert sia.image.dtype == np.int64 and np.array_equal(np.asarray(sia.image), np.asarray(ref.image))


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.float32, np.float64])
def test_property_image(dtype):
    vol = voronoi((26, 30, 66), 30, 54, np.uint16)
    sia = SpatialImageAnalysis(SpatialImage(vol, voxelsize=VS), ignoredlabels=0, return_type=DICT, background=1)
    volumes = sia.volume(real=False)
    prop = dict((l, (v % 200) + 0.5) for l, v in list(volumes.items())[::2])   # every other label has no value
    prop[1] = 99.0                                                              # the background's value is ignored
    got = sia.property_image(prop, dtype=dtype)
    want = sia_oracle.property_image(vol, prop, 1, dtype=dtype)
    assert got.dtype == np.dtype(dtype) and got.shape == vol.shape
    assert np.array_equal(np.asarray(got), want)
    assert got.voxelsize == VS


def test_c_abi_relabel_and_map_raw(gpu_ctx):
    rng = np.random.default_rng(55)
    for dtype, shape in ((np.uint16, (5, 7, 33)), (np.uint32, (3, 4, 130)), (np.uint32, (2, 3, 5))):
        vol = rng.integers(0, 50, size=shape).astype(dtype)
        lut = rng.integers(0, 60000, size=40).astype(np.uint32)        # labels 40..49 are beyond the table
        gpu_ctx.set_volume(vol)
        gpu_ctx.relabel(lut)
        out = np.empty_like(vol)
        gpu_ctx.get_volume(out)
        want = np.where(vol < 40, lut[np.minimum(vol, 39)], vol).astype(dtype)
        assert np.array_equal(out, want)
        m = gpu_ctx.map_labels(np.arange(40, dtype=np.float64) * 0.25, -1.0, vol)
        gpu_ctx.set_volume(vol)
        m = gpu_ctx.map_labels(np.arange(40, dtype=np.float64) * 0.25, -1.0, vol)
        assert np.array_equal(m, np.where(vol < 40, vol * 0.25, -1.0))
    with pytest.raises(_capi.TissueScanError) as e:
        gpu_ctx.set_volume(np.zeros((2, 2, 2), dtype=np.uint16))
        gpu_ctx.relabel(np.array([70000], dtype=np.uint32))
    assert e.value.code == _capi.TA_ERANGE

"""Method-by-method comparison of the drop-in class against the oracle's mirror of the reference
API.  Used on CPU (extraction injected from the oracle's integers) and on the GPU (real sweep)."""
import numpy as np


def same_slices(a, b):
    if a is None or b is None:
        return a is None and b is None
    return tuple((s.start, s.stop) for s in a) == tuple((s.start, s.stop) for s in b)


def close(a, b, rtol=1e-6, atol=1e-9):
    np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=rtol, atol=atol)


def compare_api(sia, ref, some_labels=None, check_inertia=True):
    """sia: tissue_analysis_amd analysis object, ref: OracleSIA on the same image/arguments."""
    labels = ref.labels()
    assert sia.labels() == labels
    assert sia.nb_labels() == ref.nb_labels()
    assert sia.is3D() and sia.background() == ref.background()
    assert sia.ignoredlabels() == ref.ignoredlabels()
    pick = some_labels if some_labels is not None else labels[:: max(1, len(labels) // 7)][:8]
    one = pick[len(pick) // 2]

    # bounding boxes: all call forms
    bb, rb = sia.boundingbox(), ref.boundingbox()
    assert sorted(bb) == sorted(rb) and all(same_slices(bb[k], rb[k]) for k in rb)
    assert same_slices(sia.boundingbox(one), ref.boundingbox(one))
    bl, rl = sia.boundingbox(pick), ref.boundingbox(pick)
    assert all(same_slices(bl[k], rl[k]) for k in pick)
    assert sia.boundingbox(one, real=True) == ref.boundingbox(one, real=True)
    assert sia.boundingbox(10 ** 6) is None and ref.boundingbox(10 ** 6) is None
    assert len(sia._bbox) == len(ref._bbox)

    # volume
    for real in (True, False):
        v, rv = sia.volume(labels, real=real), ref.volume(labels, real=real)
        assert sorted(v) == sorted(rv)
        close([v[l] for l in labels], [rv[l] for l in labels], rtol=1e-12)
    close(list(sia.volume(one).values()), list(ref.volume(one).values()))

    # barycentre
    for real in (True, False):
        c, rc = sia.center_of_mass(labels, real=real), ref.center_of_mass(labels, real=real)
        for l in labels:
            close(c[l], rc[l])
    close(sia.center_of_mass(one), ref.center_of_mass(one))

    # neighbours
    n, rn = sia.neighbors(), ref.neighbors()
    assert sorted(n) == sorted(rn)
    for k in rn:
        assert sorted(n[k]) == sorted(rn[k]), k
    assert sorted(sia.neighbors(one)) == sorted(ref.neighbors(one))
    nl, rnl = sia.neighbors(pick), ref.neighbors(pick)
    assert all(sorted(nl[k]) == sorted(rnl[k]) for k in pick)
    assert sia.neighbors_number() == ref.neighbors_number()
    assert sia.neighbors_number(one) == ref.neighbors_number(one)
    if ref.background() is not None:
        assert sorted(sia.neighbors(ref.background())) == sorted(ref.neighbors(ref.background()))

    # wall areas
    close(sia.get_voxel_face_surface(), ref.get_voxel_face_surface())
    for a, b in zip(sia.neighbor_kernels(), ref.neighbor_kernels()):
        assert np.array_equal(a, b)
    for real in (True, False):
        neigh = sorted(rn[one])
        if neigh:
            w, rw = sia.cell_wall_area(one, neigh, real), ref.cell_wall_area(one, neigh, real)
            assert sorted(w) == sorted(rw)
            close([w[k] for k in sorted(rw)], [rw[k] for k in sorted(rw)])
            close(sia.cell_wall_area(one, neigh[0], real), ref.cell_wall_area(one, neigh[0], real))
        wa, rwa = sia.wall_areas(real=real), ref.wall_areas(real=real)
        assert sorted(wa) == sorted(rwa)
        close([wa[k] for k in sorted(rwa)], [rwa[k] for k in sorted(rwa)])
    sub = dict((l, sorted(rn[l])) for l in pick)
    wa, rwa = sia.wall_areas(sub), ref.wall_areas(sub)
    assert sorted(wa) == sorted(rwa)
    close([wa[k] for k in sorted(rwa)], [rwa[k] for k in sorted(rwa)])

    # contact-area filtering
    areas = sorted(rwa.values())
    if areas:
        thr = areas[len(areas) // 2]
        f, rf = sia.neighbors(pick, min_contact_area=thr, verbose=False), ref.neighbors(pick, min_contact_area=thr)
        assert all(sorted(f[k]) == sorted(rf[k]) for k in pick)
        f1, rf1 = sia.neighbors(one, min_contact_area=thr, real_area=False, verbose=False), \
            ref.neighbors(one, min_contact_area=thr, real_area=False)
        assert sorted(f1) == sorted(rf1)

    # inertia
    if check_inertia:
        for real in (True, False):
            (vec, val), (rvec, rval) = sia.inertia_axis(pick, real=real), ref.inertia_axis(pick, real=real)
            for l in pick:
                close(val[l], rval[l])
                v, r = np.asarray(vec[l]), np.asarray(rvec[l])
                lam = np.asarray(rval[l], dtype=float)
                gap = min(abs(lam[0] - lam[1]), abs(lam[1] - lam[2])) / max(abs(lam[0]), 1e-30)
                if gap > 1e-3:
                    close(np.abs(np.sum(v * r, axis=1)), np.ones(3), atol=1e-6)
        v1, l1 = sia.inertia_axis(one)
        rv1, rl1 = ref.inertia_axis(one)
        close(l1, rl1)
        assert len(v1) == 3

    # margins and layers
    assert sorted(sia.labels_at_stack_margins()) == sorted(ref.labels_at_stack_margins())
    assert sorted(sia.labels_at_stack_margins(2)) == sorted(ref.labels_at_stack_margins(2))
    if ref.background() is not None:
        assert sorted(sia.cell_first_layer()) == sorted(ref.cell_first_layer())
        assert sorted(sia.cell_first_layer(filter_by_area=False)) == sorted(ref.cell_first_layer(filter_by_area=False))
        assert sorted(sia.cell_second_layer()) == sorted(ref.cell_second_layer())
    if len(pick) > 1:
        assert same_slices(sia.region_boundingbox(pick), ref.region_boundingbox(pick))

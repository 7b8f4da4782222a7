"""GPU tests of storm_multi (include/storm_hip.h): one scan cut over several devices inside the library -- contiguous
frequency slices + one halo point, brackets gathered in ascending order and dealt cyclically for refinement, modes dealt
cyclically for the eigenvectors.  Must be bit-identical to one storm_solver (and to the oracle).  On a one-GPU box the device
list repeats device 0, which exercises the same slicing, gather and deal; with more GPUs visible the real devices are used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


def device_lists():
    import torch
    n = torch.cuda.device_count()
    lists = [[0], [0, 0], [0, 0, 0]]
    if n >= 2:
        lists += [list(range(min(n, 8)))]
    return lists


@pytest.mark.parametrize("scheme,ell,n", [("colloc4", 2, 3001), ("magnus2", 1, 700), ("colloc2", 0, 5)])
def test_multi_equals_single_handle(oracle, oracle_models, scheme, ell, n):
    import storm_amd
    model = oracle_models("zams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    one = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    f = oracle.linspace(1.0, 25.0, n)
    ulp = oracle.relative_to_ulp(1e-8)
    ref, rdets, rinv = one.scan_and_optimize(f, ulp)
    roots = np.array([b["root"] for b in ref])
    rdet, rvec = one.eigenvector(roots) if roots.size else (np.empty(0), np.empty((0, (o.n_int + 1) * 4)))
    oref, odets, _ = o.scan_and_optimize(f, ulp, threads=16)
    assert same(rdets, odets) and [b["root"] for b in ref] == [b["root"] for b in oref]
    # both protocols (hostlogic/multi.hpp): brackets refined where they were found (default) and dealt cyclically ("deal" = 1)
    for devices, deal in [(d, k) for d in device_lists() for k in (0, 1)]:
        multi = storm_amd.HipMulti(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p, devices)
        multi.set_option("deal", deal)
        got, gdets, ginv = multi.scan_and_optimize(f, ulp)
        assert same(gdets, rdets) and ginv == rinv, devices
        assert [b["index"] for b in got] == [b["index"] for b in ref], devices
        for key in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
            assert same([b[key] for b in got], [b[key] for b in ref]), (devices, key)
        assert [b["evals"] for b in got] == [b["evals"] for b in ref]
        if roots.size:
            gdet, gvec = multi.eigenvector(roots)
            assert same(gdet, rdet) and same(gvec, rvec), devices
        st = multi.stats()
        assert st["det_evals"] >= n and st["eigenvectors"] == roots.size
        multi.close()


def test_multi_edge_cases_and_errors(oracle, oracle_models):
    import storm_amd
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc2", model.field("r_coord"))
    multi = storm_amd.HipMulti(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p, [0, 0, 0, 0])
    res, dets, inv = multi.scan_and_optimize(np.array([]), 1)
    assert res == [] and dets.size == 0 and not inv
    for n in (1, 2, 3):                              # fewer grid points than devices
        f = oracle.linspace(3.0, 4.0, n) if n > 1 else np.array([3.0])
        res, dets, inv = multi.scan_and_optimize(f, 1)
        rres, rdets, _ = o.scan_and_optimize(f, 1)
        assert same(dets, rdets) and same([b["root"] for b in res], [b["root"] for b in rres])
        assert [b["index"] for b in res] == [b["index"] for b in rres]
    # a non-finite discriminant in ONE slice: the flag is the OR over the slices (frequency 0, ell != 0: division by zero)
    f = np.concatenate([[0.0], oracle.linspace(1.0, 12.0, 40)])
    rres, _, rinv = o.scan_and_optimize(f, 1)
    res, _, inv = multi.scan_and_optimize(f, 1)
    assert inv == rinv and same([b["root"] for b in res], [b["root"] for b in rres])
    with pytest.raises(storm_amd.StormError) as e:
        multi.scan_and_optimize(oracle.linspace(1.0, 25.0, 200), 1, cap=2)
    assert e.value.code == -5
    with pytest.raises(storm_amd.StormError) as e:
        storm_amd.HipMulti(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p, [0, 99])
    assert e.value.code == -1
    multi.set_option("arithmetic", 1)                # forwarded to every solver
    got, _, _ = multi.scan_and_optimize(oracle.linspace(1.0, 12.0, 200), oracle.relative_to_ulp(1e-8))
    ref, _, _ = o.scan_and_optimize(oracle.linspace(1.0, 12.0, 200), oracle.relative_to_ulp(1e-8), threads=16)
    assert np.allclose([b["root"] for b in got], [b["root"] for b in ref], rtol=1e-10, atol=0)

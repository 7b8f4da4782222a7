"""GPU test of storm_det_batch_device_noalloc (include/storm_hip.h): the allocation-free variant of
storm_det_batch_device, whose scratch (singular-stage flag word, segment transfer matrices) is a file-scope device array.
Called through the C ABI and compared with the existing entry point's output bit for bit, on two calls in a row, the second
with a smaller input (the scratch keeps the first call's contents), in both arithmetic builds and over both kernel families
(segment path below hex_min_n in the FMA build, det_hex otherwise)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("arithmetic", [0, 1])
def test_noalloc_equals_det_batch_device_two_calls_in_a_row(oracle, oracle_models, arithmetic):
    import torch
    import storm_amd
    model = oracle_models("tams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", arithmetic)
    # (n first call, n second call): hex then segments/small, small then smaller, large then large-but-smaller
    for n1, n2 in ((3000, 200), (200, 7), (40, 1), (5000, 700)):
        f1 = torch.from_numpy(oracle.linspace(1.0, 25.0, n1)).cuda()
        f2 = torch.from_numpy(oracle.linspace(2.0, 19.0, n2) if n2 > 1 else np.array([7.25])).cuda()
        want1, want2 = torch.zeros_like(f1), torch.zeros_like(f2)
        got1, got2 = torch.full_like(f1, -1.0), torch.full_like(f2, -1.0)
        torch.cuda.synchronize()
        h.det_device(f1.data_ptr(), n1, want1.data_ptr())                    # the existing entry point (synchronous)
        h.det_device(f2.data_ptr(), n2, want2.data_ptr())
        h.det_device_noalloc(f1.data_ptr(), n1, got1.data_ptr())             # library stream: returns when done
        h.det_device_noalloc(f2.data_ptr(), n2, got2.data_ptr())             # second call, smaller input, same scratch
        assert np.array_equal(bits(got1), bits(want1)), (n1, arithmetic)
        assert np.array_equal(bits(got2), bits(want2)), (n2, arithmetic)
        # on a caller's stream: two calls back to back without a host synchronisation in between
        st = torch.cuda.Stream()
        got1.fill_(-1.0)
        got2.fill_(-1.0)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            h.det_device_noalloc(f1.data_ptr(), n1, got1.data_ptr(), stream=st.cuda_stream)
            h.det_device_noalloc(f2.data_ptr(), n2, got2.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        assert np.array_equal(bits(got1), bits(want1)) and np.array_equal(bits(got2), bits(want2))
    # batches of a few wave rounds plus a small remainder: the FMA build sends the remainder through the segment path
    # (storm_det_batch_device's choice); 4 x wave slots is 4096 .. 16384 frequencies per round
    for n1 in (12501, 16500, 33000):
        f1 = torch.from_numpy(oracle.linspace(1.0, 25.0, n1)).cuda()
        want1, got1 = torch.zeros_like(f1), torch.full_like(f1, -1.0)
        h.det_device(f1.data_ptr(), n1, want1.data_ptr())
        h.det_device_noalloc(f1.data_ptr(), n1, got1.data_ptr())
        assert np.array_equal(bits(got1), bits(want1)), (n1, arithmetic)
    if arithmetic == 0:   # and the parity build is still the oracle's bits
        f = oracle.linspace(1.0, 25.0, 200)
        d = torch.from_numpy(f).cuda()
        out = torch.zeros_like(d)
        h.det_device_noalloc(d.data_ptr(), f.size, out.data_ptr())
        assert np.array_equal(bits(out), o.det(f, threads=16).view(np.uint64))


def test_noalloc_allocates_nothing_and_rejects_oversize(oracle, oracle_models):
    import torch
    import storm_amd
    from storm_amd import ffi
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "magnus4", model.field("r_coord"))
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", 1)
    f = torch.from_numpy(oracle.linspace(1.0, 20.0, 300)).cuda()
    out = torch.zeros_like(f)
    ref = torch.zeros_like(f)
    h.det_device(f.data_ptr(), 300, ref.data_ptr())
    ffi.memory_cache_trim()
    ffi.memory_cache_limit(0, 0)             # anything the library allocated from here on would show in HIP's free memory
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        h.det_device_noalloc(f.data_ptr(), 300, out.data_ptr())
    assert torch.cuda.mem_get_info()[0] == free0
    assert np.array_equal(bits(out), bits(ref))
    ffi.memory_cache_limit(16 << 30, 2 << 30)
    # a forced segment length whose transfer matrices do not fit the scratch (1000 x (1 + 3032/16 -> 192) x 128 B > 16 MiB):
    # the plan is dropped and det_hex takes the batch -- the bits of storm_det_batch_device with "path" = 4
    big = torch.from_numpy(oracle.linspace(1.0, 20.0, 1000)).cuda()
    big_out, big_ref = torch.zeros_like(big), torch.zeros_like(big)
    h.set_option("segments", 16)
    h.set_option("hex_min_n", 2000)
    h.det_device_noalloc(big.data_ptr(), 1000, big_out.data_ptr())
    h.set_option("path", 4)
    h.det_device(big.data_ptr(), 1000, big_ref.data_ptr())
    h.set_option("path", 0)
    assert np.array_equal(bits(big_out), bits(big_ref))
    h.set_option("segments", 0)
    h.set_option("hex_min_n", 512)
    with pytest.raises(storm_amd.StormError) as e:      # beyond STORM_NOALLOC_MAX_N (rejected before anything is read)
        h.det_device_noalloc(f.data_ptr(), (1 << 24) + 1, out.data_ptr())
    assert e.value.code == -5
    h.det_device_noalloc(f.data_ptr(), 300, out.data_ptr())   # still usable
    assert np.array_equal(bits(out), bits(ref))


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


@pytest.mark.parametrize("arithmetic", [0, 1])
def test_noalloc_follows_the_path_options_and_small_grids(oracle, oracle_models, arithmetic):
    """The kernel choice is storm_det_batch_device's, options included; where that one would stage step matrices ("path" =
    2; a small batch with no segment plan: parity build, or an FMA-build grid of fewer than 64 intervals) det_hex stands
    in: the same bits in the parity build (every kernel path is bit-identical there), the FMA build's tolerance otherwise."""
    import torch
    import storm_amd
    model = oracle_models("zams")
    for grid in (model.field("r_coord"), model.field("r_coord")[::60]):
        o = oracle.Solver(model, 2, 0, "colloc6", grid)
        assert (o.n_int < 64) == (grid.size < 100)
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        h.set_option("arithmetic", arithmetic)
        for n in (100, 777):
            f = torch.from_numpy(oracle.linspace(1.0, 20.0, n)).cuda()
            for path in (0, 1, 2, 3, 4):
                h.set_option("path", path)
                want, got = torch.zeros_like(f), torch.full_like(f, -1.0)
                h.det_device(f.data_ptr(), n, want.data_ptr())
                h.det_device_noalloc(f.data_ptr(), n, got.data_ptr())
                w, g = want.cpu().numpy(), got.cpu().numpy()
                staged = path == 2 or (path == 0 and n < 512 and (arithmetic == 0 or o.n_int < 64))
                if arithmetic == 0 or not staged:
                    assert np.array_equal(bits(got), bits(want)), (o.n_int, n, path)
                else:
                    assert np.all(np.abs(g - w) <= 1e-9 * local_scale(w)), (o.n_int, n, path)
                    assert np.array_equal(np.signbit(g), np.signbit(w))

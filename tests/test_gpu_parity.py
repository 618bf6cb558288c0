"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Needs a GPU.

Bars.  The claim is the strong one: phi BIT-IDENTICAL to the O(D) oracle
(oracle/fast_oracle.c), which evaluates the same prefix/suffix-min formulas as the kernels
and is itself checked against the faithful oracle on the CPU.  Beside it, against the
faithful checkerboard oracle (oracle/diffusion_oracle.c, the reference's operation order)
wherever brute force is affordable: phi within 1e-9 (f64) / 1e-5 (f32) relative, both
norm-wise (max|d| / max(1, max|phi|)) and element-wise (|d| <= tol * max(1, |phi_ref|) at
every entry).  Cost volume bit-exact; labels identical; energy 1e-9.
"""
import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import pyoracle
from oracle.pyoracle import FastOracle, OracleGraph
from tests.util import elem_err, random_pair, random_phi, rel_err

pytestmark = pytest.mark.gpu

load_package()
from stereo_vision_amd import diffstereo, synth  # noqa: E402

TOL = {"f64": 1e-9, "f32": 1e-5}

# (H, W, D): odd sizes, W < D, D not a multiple of 4, every lanes-per-pixel variant
SMALL = [(1, 1, 1), (1, 2, 2), (2, 1, 2), (1, 7, 3), (7, 1, 3), (2, 2, 2), (3, 3, 1), (5, 5, 40),
         (4, 7, 3), (6, 9, 5), (7, 12, 16), (5, 33, 40), (9, 8, 13), (3, 70, 64), (4, 37, 65),
         (3, 50, 128), (2, 21, 129), (3, 40, 200), (2, 300, 256),
         # one pixel per wavefront: whole iterations run in the fused kernel (strips of 30 / 14
         # columns, row segments, interior fast path from column 256 on)
         (20, 400, 256), (70, 330, 200), (5, 291, 256), (33, 61, 130), (1, 350, 256), (9, 31, 256)]


@pytest.mark.parametrize("H,W,D", SMALL)
def test_cost_volume_bit_exact(H, W, D):
    left, right = random_pair(H, W, seed=H + 31 * W + 977 * D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2)
    assert np.array_equal(g.get_cost(), pyoracle.cost_volume(left, right, D))
    g.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("H,W,D", SMALL)
def test_phi_matches_oracles(H, W, D, dtype):
    left, right = random_pair(H, W, seed=H + 31 * W + 977 * D)
    phi0 = random_phi(H, W, D, dtype, seed=5)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    g.set_phi(phi0)
    assert np.array_equal(g.get_phi(), phi0), "set/get round trip"
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    f.potentials[...] = phi0
    brute = H * W * D * D <= 400_000
    if brute:
        o = OracleGraph(left, right, D, 1.2, dtype=dtype)
        o.set_potentials(phi0)
    for it in range(3):
        g.iterate(1)
        f.iterate(1)
        phi = g.get_phi()
        assert np.array_equal(phi, f.potentials), "bit-exact vs O(D) oracle, iteration %d" % it
        if brute:
            o.diffusion_act_checkerboard()
            assert rel_err(phi, o.potentials) <= TOL[dtype], "iteration %d" % it
            assert elem_err(phi, o.potentials) <= TOL[dtype], "element-wise, iteration %d" % it
    e_gpu, resid = g.energy(with_residual=True)
    e_cpu = f.energy()
    assert abs(e_gpu - e_cpu) <= (1e-9 if dtype == "f64" else 1e-5) * max(1.0, abs(e_cpu))
    if H * W > 1:
        assert resid <= (1e-9 if dtype == "f64" else 1e-3)
    assert np.array_equal(g.labels(0), f.labels(0))
    assert np.array_equal(g.labels(1), f.labels(1))
    assert g.checksum() == diffstereo.host_checksum(phi, 0, H, W, D)
    g.close()


def test_bundled_pair_100_iterations(golden_dir):
    """The reference's own images (images/left.pgm, right.pgm; D = 40, lambda = 1.2 as in
    diffusion_stereo/src/main.rs:44-47), 100 iterations, against the faithful
    checkerboard oracle and the committed golden phi."""
    left, _, _ = pyoracle.pgm_reader(golden_dir + "/left.pgm")
    right, _, _ = pyoracle.pgm_reader(golden_dir + "/right.pgm")
    gold = np.load(golden_dir + "/bundled_cb_100.npz")
    for dtype in ("f64", "f32"):
        g = diffstereo.DiffusionGraph(left, right, 40, 1.2, dtype=dtype)
        g.iterate(100)
        phi = g.get_phi()
        assert rel_err(phi, gold["phi_f64"]) <= TOL[dtype]
        assert abs(g.energy() - float(gold["energy_f64"])) <= 1e-3 * max(1.0, abs(float(gold["energy_f64"])))
        g.close()


def test_medium_bit_exact_and_properties():
    """A size the brute-force oracle cannot reach: bit-exact vs the O(D) oracle, zero
    vertex residual after whole iterations, non-decreasing dual energy."""
    H, W, D = 48, 320, 128
    left, right, _ = synth.synthetic_pair(H, W, D)
    for dtype in ("f32", "f64"):
        g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
        f = FastOracle(left, right, D, 1.2, dtype=dtype)
        prev = g.energy()
        for _ in range(4):
            g.iterate(1)
            f.iterate(1)
            e, resid = g.energy(with_residual=True)
            assert e >= prev - 1e-6 * max(1.0, abs(prev))
            assert resid <= (1e-9 if dtype == "f64" else 1e-3)
            prev = e
        assert np.array_equal(g.get_phi(), f.potentials)
        assert np.array_equal(g.labels(0), f.labels(0))
        g.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_bands_bit_identical_to_whole_image(dtype):
    """Three bands on one GPU with the halo rows moved through ds_halo's device
    pointers must reproduce the unpartitioned result bit for bit."""
    import ctypes as C
    hip = diffstereo.hip_runtime()
    H, W, D = 11, 37, 20
    left, right = random_pair(H, W, seed=77)
    phi0 = random_phi(H, W, D, dtype, seed=3)
    whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    whole.set_phi(phi0)
    cuts = [(0, 4), (4, 5), (5, 11)]
    bands = [diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype, row_begin=a, row_end=b)
             for a, b in cuts]
    for b, (a, e) in zip(bands, cuts):
        lo, hi = max(a - 1, 0), min(e + 1, H)
        b.set_phi(phi0[lo:hi], lo, hi)  # own rows + halo rows

    def exchange(colour):
        for up, down in zip(bands[:-1], bands[1:]):
            for src, dst in (((up, 1, 0), (down, 0, 1)), ((down, 0, 0), (up, 1, 1))):
                sp, n = src[0].halo(colour, src[1], src[2])
                dp, m = dst[0].halo(colour, dst[1], dst[2])
                assert n == m
                src[0].synchronize()
                assert hip.hipMemcpy(C.c_void_p(dp), C.c_void_p(sp), C.c_size_t(n), 3) == 0

    for _ in range(3):
        for colour in (0, 1):
            whole.sweep(colour)
            for b in bands:
                b.sweep(colour)
            exchange(colour)
    whole.synchronize()
    ref = whole.get_phi()
    for b, (a, e) in zip(bands, cuts):
        assert np.array_equal(b.get_phi(), ref[a:e])
    assert abs(sum(b.energy() for b in bands) - whole.energy()) <= 1e-9 * max(1.0, abs(whole.energy()))
    assert sum(b.checksum() for b in bands) % 2**64 == whole.checksum()
    lab = np.concatenate([b.labels(0) for b in bands], axis=0)
    assert np.array_equal(lab, whole.labels(0))
    for b in bands + [whole]:
        b.close()


def test_error_paths():
    left, right = random_pair(4, 4)
    with pytest.raises(diffstereo.DiffStereoError):
        diffstereo.DiffusionGraph(left, right, 0, 1.2)
    with pytest.raises(diffstereo.DiffStereoError):
        diffstereo.DiffusionGraph(left, right, 257, 1.2)
    with pytest.raises(diffstereo.DiffStereoError):
        diffstereo.DiffusionGraph(left, right, 4, float("nan"))
    band = diffstereo.DiffusionGraph(left, right, 4, 1.2, row_begin=1, row_end=3)
    with pytest.raises(diffstereo.DiffStereoError):
        band.iterate(1)  # a band needs the halo exchange
    with pytest.raises(diffstereo.DiffStereoError):
        band.sweep(0, 0, 2)  # outside the band
    band.close()


def test_placement_does_not_change_results(monkeypatch):
    """DS_PLACEMENT=search (phi in two separately allocated plane groups, placements timed at
    ds_create) must give the same bits as the default, one plain allocation."""
    H, W, D = 258, 2048, 256
    left, right, _ = synth.synthetic_pair(H, W, D)
    a = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    info = a.placement_info()
    assert info == {"tries": 0, "probe_ms": 0.0, "split": False}  # the default: one plain allocation
    monkeypatch.setenv("DS_PLACEMENT", "search")
    monkeypatch.setenv("DS_PLACEMENT_TRIES", "2")
    c = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    info = c.placement_info()
    assert info["split"] and 1 <= info["tries"] <= 2 and info["probe_ms"] > 0
    assert c.checksum() == 0  # probing leaves phi in its initial state
    c.iterate(3)
    monkeypatch.setenv("DS_PLACEMENT", "off")  # any other value: the default
    b = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    assert b.placement_info() == {"tries": 0, "probe_ms": 0.0, "split": False}
    assert a.checksum() == b.checksum() == 0
    a.iterate(3)
    b.iterate(3)
    assert a.checksum() == b.checksum() == c.checksum()
    c.close()
    assert np.array_equal(a.labels(0), b.labels(0))
    ea, eb = a.energy(), b.energy()
    assert ea == eb
    assert np.array_equal(a.get_phi(100, 103), b.get_phi(100, 103))
    f = FastOracle(left[:40], right[:40], D, 1.2, dtype="f32")  # top rows only: same neighbourhoods
    f.iterate(3)
    # rows 0..35 of the 40-row crop are unaffected by the crop's bottom edge after 3 iterations
    assert np.array_equal(a.get_phi(0, 30), f.potentials[:30])
    a.close()
    b.close()


def test_full_size_4k_256_properties(monkeypatch):
    """The benchmark's own size (3840x2160, D=256, f32): properties that need no full-size
    oracle -- non-decreasing dual energy, zero vertex residual, two half-image bands equal to
    the whole image (checksum of checksums, energy, labels), the top rows bit-exact against
    the O(D) oracle on a crop, crossing-out idempotent."""
    import ctypes as C
    hip = diffstereo.hip_runtime()
    H, W, D = 2160, 3840, 256
    monkeypatch.setenv("DS_PLACEMENT_TRIES", "2")
    left, right, _ = synth.synthetic_pair(H, W, D)
    whole = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    cuts = [(0, 1080), (1080, 2160)]
    bands = [diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32", row_begin=a, row_end=b) for a, b in cuts]
    up, down = bands

    def exchange(colour):
        for src, dst in (((up, 1, 0), (down, 0, 1)), ((down, 0, 0), (up, 1, 1))):
            sp, n = src[0].halo(colour, src[1], src[2])
            dp, m = dst[0].halo(colour, dst[1], dst[2])
            assert n == m
            src[0].synchronize()
            assert hip.hipMemcpy(C.c_void_p(dp), C.c_void_p(sp), C.c_size_t(n), 3) == 0

    prev = whole.energy()
    for _ in range(3):
        for colour in (0, 1):
            whole.sweep(colour)
            for b in bands:
                b.sweep(colour)
            exchange(colour)
        e, resid = whole.energy(with_residual=True)
        assert e >= prev - 1e-6 * abs(prev) and resid <= 1e-3
        prev = e
    assert (up.checksum() + down.checksum()) % 2**64 == whole.checksum()
    # the same three iterations through ds_iterate (whole iterations in the fused kernel)
    fused = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    assert fused.fused_rows() == (0, H)
    fused.iterate(3)
    assert fused.checksum() == whole.checksum()
    assert fused.energy() == prev
    fused.close()
    e_bands = up.energy() + down.energy()
    assert abs(e_bands - prev) <= 1e-9 * abs(prev)
    lab = whole.labels(0)
    assert np.array_equal(np.concatenate([up.labels(0), down.labels(0)]), lab)
    assert (lab <= np.minimum(np.arange(W)[None, :], D - 1)).all()
    f = FastOracle(left[:40], right[:40], D, 1.2, dtype="f32")
    f.iterate(3)
    assert np.array_equal(whole.get_phi(0, 30), f.potentials[:30])
    assert np.array_equal(lab[:30], f.labels(0)[:30])
    for b in bands:
        b.close()
    r1 = whole.crossing_out(5.0)
    r2 = whole.crossing_out(5.0)
    assert np.array_equal(r1["labels"], r2["labels"]) and r1["not_empty"] == r2["not_empty"]
    whole.close()


def test_1080p_128_f32_properties(monkeypatch):
    """Synthetic 1920x1080 pair, 128 disparities, f32 phi, whole iterations through ds_iterate:
    the fused kernel's two-pixels-per-wavefront variant at full size.  Same bits as the
    half-sweep path (checksum, energy), top rows bit-exact against the O(D) oracle on a crop,
    non-decreasing dual energy, zero vertex residual, labels inside the valid triangle."""
    H, W, D = 1080, 1920, 128
    monkeypatch.setenv("DS_PLACEMENT_TRIES", "2")
    left, right, _ = synth.synthetic_pair(H, W, D)
    fused = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    assert fused.fused_rows() == (0, H)
    monkeypatch.setenv("DS_FUSED", "off")
    swept = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f32")
    monkeypatch.delenv("DS_FUSED")
    assert swept.fused_rows() == (0, 0)
    prev = fused.energy()
    for _ in range(4):
        fused.iterate(1)
        swept.iterate(1)
        e, resid = fused.energy(with_residual=True)
        assert e >= prev - 1e-6 * abs(prev) and resid <= 1e-3
        assert e == swept.energy()
        prev = e
    fused.iterate(6)  # several iterations in one call: the first colour is materialised once
    swept.iterate(6)
    assert fused.checksum() == swept.checksum()
    assert fused.energy() == swept.energy() >= prev
    f = FastOracle(left[:48], right[:48], D, 1.2, dtype="f32")
    f.iterate(10)
    # after 10 iterations rows 0..27 of the 48-row crop have not seen the crop's bottom edge
    assert np.array_equal(fused.get_phi(0, 24), f.potentials[:24])
    lab = fused.labels(0)
    assert np.array_equal(lab, swept.labels(0))
    assert np.array_equal(lab[:24], f.labels(0)[:24])
    assert (lab <= np.minimum(np.arange(W)[None, :], D - 1)).all()
    fused.close()
    swept.close()


def test_8k_256_f64_band_of_eight(monkeypatch):
    """One of the eight row bands of the 7680x4320, 256-disparity, f64 workload (rows 0..540,
    51 GB of planes): the band protocol around the fused kernel (boundary rows through
    half-sweeps, interior in one kernel) against plain half-sweeps on a second handle of the
    same band.  No neighbour is attached, so both see the same (initial) bottom halo and must
    agree bit for bit; the top rows are bit-exact against the O(D) oracle on a crop."""
    H, W, D = 4320, 7680, 256
    rb, re = 0, 540
    monkeypatch.setenv("DS_PLACEMENT_TRIES", "2")
    left, right, _ = synth.synthetic_pair(H, W, D)
    band = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64", row_begin=rb, row_end=re)
    fb, fe = band.fused_rows()
    assert (fb, fe) == (0, re - 1)
    monkeypatch.setenv("DS_FUSED", "off")
    plain = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64", row_begin=rb, row_end=re)
    monkeypatch.delenv("DS_FUSED")
    iters = 3
    for _ in range(iters):
        band.sweep(0, re - 2, re)    # first colour, the two rows next to the neighbour
        band.sweep_next(re - 1, re)  # second colour, the last row, into the upcoming planes
        band.iterate_rows(fb, fe)
        plain.sweep(0)
        plain.sweep(1)
    band.finish_iterations()
    band.synchronize()
    plain.synchronize()
    assert band.checksum() == plain.checksum()
    e, resid = band.energy(with_residual=True)
    assert e == plain.energy() and resid <= 1e-9
    assert np.array_equal(band.get_phi(re - 3, re), plain.get_phi(re - 3, re))
    f = FastOracle(left[:40], right[:40], D, 1.2, dtype="f64")
    f.iterate(iters)
    assert np.array_equal(band.get_phi(0, 30), f.potentials[:30])
    assert np.array_equal(band.labels(0)[:30], f.labels(0)[:30])
    band.close()
    plain.close()


def test_8k_wide_f64_band_bit_exact():
    """The widest configuration's row width (7680, D=256, f64) on a 64-row image: bit-exact
    against the O(D) oracle, and the raster-order sweep agrees with it on the dual optimum's
    lower bound ordering (energy grows under both)."""
    H, W, D = 64, 7680, 256
    left, right, _ = synth.synthetic_pair(H, W, D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype="f64")
    f = FastOracle(left, right, D, 1.2, dtype="f64")
    g.iterate(2)
    f.iterate(2)
    assert np.array_equal(g.get_phi(), f.potentials)
    assert abs(g.energy() - f.energy()) <= 1e-9 * abs(f.energy())
    assert np.array_equal(g.labels(0), f.labels(0))
    assert g.checksum() == diffstereo.host_checksum(f.potentials, 0, H, W, D)
    g.close()


@pytest.mark.parametrize("waves,dtype,shape", [("8", "f32", (20, 400, 256)), ("4", "f32", (9, 150, 100)),
                                               ("4", "f64", (12, 300, 256)), ("off", "f32", (20, 400, 256))])
def test_fused_kernel_block_sizes_and_half_sweep_fallback(monkeypatch, waves, dtype, shape):
    """The other block sizes of the fused iteration kernel (DS_FUSED_WAVES) and the half-sweep
    path (DS_FUSED=off) give the same bits as the default."""
    H, W, D = shape
    if waves == "off":
        monkeypatch.setenv("DS_FUSED", "off")
    else:
        monkeypatch.setenv("DS_FUSED_WAVES", waves)
    left, right = random_pair(H, W, seed=H + W + D)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    assert (g.fused_rows() == (0, H)) == (waves != "off")
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    g.iterate(3)
    f.iterate(3)
    assert np.array_equal(g.get_phi(), f.potentials)
    assert g.checksum() == diffstereo.host_checksum(f.potentials, 0, H, W, D)
    g.close()


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        D = int(rng.choice([rng.integers(1, 17), rng.integers(17, 65), rng.integers(65, 129), rng.integers(129, 257)]))
        W = int(rng.choice([rng.integers(1, 12), rng.integers(12, 130), rng.integers(130, 420)]))
        H = int(rng.integers(1, 36))
        out.append((H, W, D, "f64" if rng.integers(0, 2) else "f32"))
    return out


@pytest.mark.parametrize("H,W,D,dtype", _random_shapes(48, seed=20240611))
def test_random_shapes_bit_exact(H, W, D, dtype):
    """Randomly drawn shapes (all pixel-per-wavefront variants, strips cut by the image edge,
    single rows and columns, W < D): two iterations through ds_iterate against the O(D) oracle."""
    left, right = random_pair(H, W, seed=H * 1000 + W)
    g = diffstereo.DiffusionGraph(left, right, D, 1.2, dtype=dtype)
    f = FastOracle(left, right, D, 1.2, dtype=dtype)
    g.iterate(2)
    f.iterate(2)
    assert np.array_equal(g.get_phi(), f.potentials)
    assert np.array_equal(g.labels(0), f.labels(0))
    g.close()

"""Properties of the extended renderer's sampling, checked on the CPU oracle (no GPU):

 * the counter-based keys: reproducible, independent of evaluation order, uniform sub-pixel
   offsets, different per pixel and per sample;
 * the unit-vector and cos/sin building blocks against numpy;
 * unbiasedness: a jittered 256-spp image converges to the image obtained by regular 16x16
   super-sampling of the reference's own pixel kernel.
"""
import ctypes

import numpy as np
import pytest

import oracle


def test_keys_are_a_pure_function_of_seed_pixel_sample():
    L = oracle.lib()
    a = [L.orc_path_key(7, p, s) for p in range(50) for s in range(20)]
    b = [L.orc_path_key(7, p, s) for p in reversed(range(50)) for s in reversed(range(20))]
    assert a == list(reversed(b))
    assert len(set(a)) > 0.99 * len(a)
    assert L.orc_path_key(7, 3, 4) == (L.orc_pixel_term(7, 3) + L.orc_sample_term(7, 4)) & 0xFFFFFFFF
    assert L.orc_path_key(8, 3, 4) != L.orc_path_key(7, 3, 4)


def sample_offsets(spp, pixel, seed=1):
    import ctypes

    L = oracle.lib()
    term = L.orc_pixel_term(seed, pixel)
    jx, jy = ctypes.c_float(), ctypes.c_float()
    out = np.zeros((spp, 2))
    for s in range(spp):
        L.orc_sample_offset(spp, s, term, ctypes.byref(jx), ctypes.byref(jy))
        out[s] = (jx.value, jy.value)
    return out


@pytest.mark.parametrize("spp", [1, 2, 3, 4, 5, 16, 64, 70, 256, 1000, 1024])
def test_sample_positions_are_one_per_cell_of_a_grid_over_the_pixel(spp):
    """The pixel's samples sit one per cell of a grid: rows of 2^floor(log2(spp)/2) cells, a shorter
    last row at its own cell width, row heights in proportion to the rows' sample counts; every
    sample at the pixel's own offset inside its cell."""
    L = oracle.lib()
    nx = 1 << L.orc_sample_row_length_log2(spp)
    assert nx * nx <= spp < 4 * nx * nx
    for pixel in (0, 7, 123456):
        o = sample_offsets(spp, pixel)
        assert np.all(o >= 0.0) and np.all(o < 1.0)
        term = L.orc_pixel_term(1, pixel)
        rx, ry = (term >> 16) / 65536.0, (term & 0xFFFF) / 65536.0
        for s in range(spp):
            r, i = divmod(s, nx)
            n_r = min(nx, spp - r * nx)
            assert abs(o[s, 0] - (i + rx) / n_r) < 1e-6
            assert abs(o[s, 1] - (r * nx + ry * n_r) / spp) < 1e-6
        # the cells tile the pixel: cell areas (1 / n_r) x (n_r / spp) are all 1 / spp


def test_sample_positions_are_uniform_over_pixels():
    """Over many pixels (their offsets are the halves of a hash) a given cell's sample is uniform
    in the cell, so all samples together are uniform in the pixel."""
    o = np.array([sample_offsets(1, p)[0] for p in range(8192)])      # one cell: the offset itself
    for j in (o[:, 0], o[:, 1]):
        assert abs(j.mean() - 0.5) < 0.01
        assert abs(j.var() - 1.0 / 12.0) < 0.004
        hist, _ = np.histogram(j, bins=16, range=(0, 1))
        expected = len(j) / 16
        chi2 = float(((hist - expected) ** 2 / expected).sum())
        assert chi2 < 45.0                      # 15 degrees of freedom, p ~ 1e-4
    # the offset inside the cell: x and y parts uncorrelated, pixels do not share offsets
    assert abs(np.corrcoef(o[:, 0], o[:, 1])[0, 1]) < 0.05
    assert len({tuple(v) for v in o.tolist()}) > 8000


def test_generator_steps_like_pcg32():
    L = oracle.lib()
    state = ctypes.c_uint32(12345)
    first = L.orc_rng_next(ctypes.byref(state))
    assert first == L.orc_pcg_hash(12345)
    assert state.value == (12345 * 747796405 + 2891336453) & 0xFFFFFFFF


def test_cossin_matches_numpy():
    L = oracle.lib()
    c, s = ctypes.c_float(), ctypes.c_float()
    worst = 0.0
    for u in np.linspace(0.0, 1.0, 4097, endpoint=False):
        L.orc_cossin_2pi(ctypes.c_float(u), ctypes.byref(c), ctypes.byref(s))
        worst = max(worst, abs(c.value - np.cos(2 * np.pi * np.float32(u))), abs(s.value - np.sin(2 * np.pi * np.float32(u))))
    assert worst < 1e-6


def test_unit_vectors_are_unit_and_cover_the_sphere():
    L = oracle.lib()
    out = (ctypes.c_float * 3)()
    rng = np.random.default_rng(3)
    v = np.empty((4000, 3))
    for i, bits in enumerate(rng.integers(0, 2 ** 32, 4000, dtype=np.uint64)):
        L.orc_unit_vector(int(bits), out)
        v[i] = out[:]
    assert np.max(np.abs(np.linalg.norm(v, axis=1) - 1.0)) < 2e-6
    assert np.all(np.abs(v.mean(axis=0)) < 0.05)
    assert np.all(np.abs((v ** 2).mean(axis=0) - 1.0 / 3.0) < 0.03)


def test_jittered_samples_converge_to_regular_supersampling():
    """Ground truth: the reference's pixel kernel at 16x the resolution (its pixel centres are a
    regular 16x16 grid inside each coarse pixel), box-filtered.  The sphere's colour varies
    smoothly, so only silhouette pixels carry sampling noise."""
    w, h, k = 96, 54, 16
    _, fine = oracle.trace_primary(w * k, h * k, want_rgba=True)
    # the row gradient of a miss is taken per COARSE row by the jittered render; compare on
    # hit coverage and colour only where the fine image is all-hit or part-hit
    fine_rgb = fine[..., :3].reshape(h, k, w, k, 3)
    truth = fine_rgb.mean(axis=(1, 3))
    _, est, _, _ = oracle.trace_paths(w, h, spp=256, max_depth=1, seed=5, jitter=True, threads=8, want_rgba=True)
    _, coarse = oracle.trace_primary(w, h, want_rgba=True)
    inside = np.zeros((h, w), dtype=bool)
    # pixels whose 16x16 fine samples all hit the sphere: no background involved
    bg_fine = oracle.trace_primary(w * k, h * k)
    row_bg = bg_fine[:, 0]                                   # first column is always background
    hit_fine = bg_fine != row_bg[:, None]
    inside = hit_fine.reshape(h, k, w, k).all(axis=(1, 3))
    assert inside.sum() > 300
    err = est[..., :3][inside] - truth[inside]
    assert abs(float(err.mean())) < 2e-4                      # unbiased
    assert float(np.sqrt((err ** 2).mean())) < 2e-3           # and converged
    # silhouette pixels: coverage-weighted mix, noisier but still unbiased
    edge = hit_fine.reshape(h, k, w, k).any(axis=(1, 3)) & ~inside
    cover_truth = hit_fine.reshape(h, k, w, k).mean(axis=(1, 3))[edge]
    # recover the estimate's coverage from the red channel: background red vs sphere red differ,
    # so just bound the colour error instead
    e2 = est[..., :3][edge] - (truth[edge])
    # truth for edge pixels mixes fine-row backgrounds; the jittered render uses the coarse row's
    # background, a difference of at most one coarse row of gradient
    grad_step = float(np.abs(coarse[1, 0, :3] - coarse[0, 0, :3]).max())
    assert float(np.abs(e2.mean())) < 0.01 + grad_step
    assert cover_truth.min() > 0.0

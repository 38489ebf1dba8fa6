"""er_denoise fills the DENOISE plane (never written by renderingKernel) with an a-trous
wavelet filter of BEAUTY guided by NORMAL.  The reference's `get_pass denoise` runs OIDN on the host, a neural
filter with no arithmetic to match, so there is no parity claim against it; what is pinned here is the filter
itself: a numpy replay of the same IEEE float32 operations, bit for bit, and its sanity (finite, alpha kept, noise
down, constant images unchanged)."""
import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes

pytestmark = pytest.mark.gpu
f32 = np.float32


def atrous_numpy(beauty, normal, levels, sigma):
    h, w = beauty.shape[:2]
    kern = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16], f32)
    src = beauty.copy()
    ys, xs = np.mgrid[0:h, 0:w]
    none = (normal[..., :3] == 0).all(-1)
    for k in range(levels):
        step = 1 << k
        kc = f32(f32(1.0) / f32(f32(sigma) * f32(sigma)) * f32(1 << k))
        acc = np.zeros((h, w, 3), f32)
        sw = np.zeros((h, w), f32)
        c = src[..., :3]
        for j in range(-2, 3):
            for i in range(-2, 3):
                qx = np.clip(xs + i * step, 0, w - 1)
                qy = np.clip(ys + j * step, 0, h - 1)
                cq = src[qy, qx, :3]
                nq = normal[qy, qx, :3]
                d = (c - cq).astype(f32)
                d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(f32) + d[..., 2] * d[..., 2]
                wc = f32(1.0) / (f32(1.0) + kc * d2)
                nd = (normal[..., 0] * nq[..., 0] + normal[..., 1] * nq[..., 1]).astype(f32) + normal[..., 2] * nq[..., 2]
                nd = np.where(none & none[qy, qx], f32(1.0), np.where(nd < 0, f32(0.0), nd)).astype(f32)
                wgt = ((kern[i + 2] * kern[j + 2]) * wc).astype(f32) * (nd * nd).astype(f32)
                acc = acc + cq * wgt[..., None]
                sw = sw + wgt
        out = src.copy()
        out[..., :3] = acc / sw[..., None]
        src = out
    return src


@pytest.mark.parametrize("levels,sigma", [(1, 1.0), (3, 0.5), (5, 1.0)])
def test_denoise_matches_numpy_replay(levels, sigma):
    sc = scenes.soup(4000, 70, 45, seed=3, hdri_size=(64, 32))
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8))
    rm.start_rendering(sc)
    rm.render(4)
    before = rm.get_pass("denoise")
    assert (before[..., :3] == 0).all() and (before[..., 3] == 1).all()          # setupKernel's values, untouched by rendering
    rm.denoise(levels, sigma)
    got = rm.get_pass("denoise")
    beauty, normal = rm.get_pass("beauty"), rm.get_pass("normal")
    rm.close()
    want = atrous_numpy(beauty, normal, levels, sigma)
    assert (got.view(np.uint32) == want.view(np.uint32)).all(), np.abs(got - want).max()
    assert np.isfinite(got).all() and (got[..., 3] == beauty[..., 3]).all()


def test_denoise_reduces_noise_and_keeps_flat_images():
    sc = scenes.cornell(96, 96)
    rm = render.RenderingManager(render.RenderParameters(max_bounces=5))
    rm.start_rendering(sc)
    rm.render(4)
    noisy = rm.get_pass("beauty")
    rm.denoise()
    den = rm.get_pass("denoise")
    rm.render(252)
    ref = rm.get_pass("beauty")          # 256 spp of the same pixels
    rm.close()
    err_noisy = np.abs(noisy[..., :3] * 5 / 4 - ref[..., :3] * 257 / 256).mean()   # undo the first-sample half weight (Appendix A)
    err_den = np.abs(den[..., :3] * 5 / 4 - ref[..., :3] * 257 / 256).mean()
    print("mean abs error vs 256 spp: 4 spp", err_noisy, "denoised", err_den)
    assert err_den < 0.7 * err_noisy

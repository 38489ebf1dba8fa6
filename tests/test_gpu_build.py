"""the device-side BVH builder (er_gpu_build.hip; since round 5 the host builder's binned SAH, level by level on
the GPU, and the default for scenes of >= 20 000 triangles; ER_FLAG_GPU_BUILD / ER_FLAG_HOST_BUILD force one).

The contract of an acceleration structure is "same nearest hit", so the image must not depend on who built the
tree: the device build against the host build, bit for bit, and against the oracle."""
import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes
from test_gpu_parity import compare, gpu_render, oracle_render

pytestmark = pytest.mark.gpu


def _accel(scene, flags):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=flags))
    rm.start_rendering(scene)
    a = rm.accel_info()
    rm.close()
    return a


@pytest.mark.parametrize("kind", ["soup", "blobs", "cornell", "torture"])
def test_device_built_bvh_gives_the_same_image(kind):
    if kind == "soup":
        sc = scenes.soup(30000, 120, 88, seed=8, hdri_size=(128, 64))
    elif kind == "blobs":
        sc = scenes.blob_instances(n_instances=30, tris_per_blob=300, x_res=96, y_res=72, grid=(5, 3, 2), spacing=0.45)
    elif kind == "cornell":
        sc = scenes.cornell(64, 48)
    else:
        sc = scenes.torture(6000, 80, 60, seed=6, n_materials=8, tex_size=16, hdri_size=(64, 32))
    assert _accel(sc, abi.FLAG_GPU_BUILD)["builder"] == 1 and _accel(sc, abi.FLAG_HOST_BUILD)["builder"] == 0
    assert _accel(sc, 0)["builder"] == (1 if sc.tri_count >= 20000 else 0)        # the default goes by the triangle count
    for sched in (abi.FLAG_WAVEFRONT, abi.FLAG_MEGAKERNEL, abi.FLAG_STREAM):
        h = gpu_render(sc, 4, max_bounces=8, flags=sched | abi.FLAG_HOST_BUILD)
        d = gpu_render(sc, 4, max_bounces=8, flags=sched | abi.FLAG_GPU_BUILD)
        for p in ("beauty", "normal", "tangent", "bitangent"):
            same = (h[p].view(np.uint32) == d[p].view(np.uint32)).all(-1)
            # only an exact distance tie between two triangles may be resolved differently by a different tree
            assert same.mean() >= 0.9995, (kind, sched, p, float(same.mean()))
        assert (h["samples"] == d["samples"]).all()


def test_device_built_bvh_against_oracle(oracle_mod):
    sc = scenes.soup(8000, 96, 64, seed=17, hdri_size=(64, 32))
    g = gpu_render(sc, 3, max_bounces=8, flags=abi.FLAG_GPU_BUILD)
    o = oracle_render(oracle_mod, sc, 3, max_bounces=8)
    compare(g, o, what="device-built BVH vs oracle")
    assert g["counters"]["bounce_samples"] == o["counters"]["bounce_samples"]


def test_device_builder_full_size():
    """1M triangles: the device build is several times faster than the host build, makes a tree of the same size (it is the same
    algorithm: node counts may differ by float rounding of the bins only), and the window of tiles rendered through it equals the one
    rendered through the host-built tree."""
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    dev, host = _accel(sc, 0), _accel(sc, abi.FLAG_HOST_BUILD)
    print(f"build: device {dev['build_ms']:.0f} ms ({dev['node_count']} wide nodes, depth {dev['max_depth']}), "
          f"host {host['build_ms']:.0f} ms ({host['node_count']} wide nodes, depth {host['max_depth']})")
    assert dev["builder"] == 1 and host["builder"] == 0
    assert abs(int(dev["node_count"]) - int(host["node_count"])) <= host["node_count"] // 100 and abs(int(dev["leaf_count"]) - int(host["leaf_count"])) <= host["leaf_count"] // 100
    a = gpu_render(sc, 2, max_bounces=8, rank=3, world=64, flags=abi.FLAG_HOST_BUILD)
    b = gpu_render(sc, 2, max_bounces=8, rank=3, world=64)
    same = (a["beauty"].view(np.uint32) == b["beauty"].view(np.uint32)).all(-1)
    assert same.mean() >= 0.9995
    assert a["counters"]["paths"] == b["counters"]["paths"]


def test_device_builder_on_clustered_geometry():
    """Tight far-apart clusters (most bins of the top splits are empty): same image as the host-built tree."""
    sc = scenes.soup(6000, 96, 72, seed=31, hdri_size=(64, 32))
    v = sc.vertices.reshape(-1, 3, 3).copy()
    c = v.mean(1, keepdims=True)
    cluster = (np.arange(len(v)) % 3)[:, None, None]
    centre = np.array([[-0.9, 0.3, 2.2], [0.8, -0.4, 3.6], [0.1, 0.7, 2.9]], np.float32)[cluster[:, 0, 0]][:, None, :]
    v = centre + (v - c) * 0.2 + (c - c.mean(0)) * 1e-3         # each cluster ~2 mm across
    sc.vertices = np.ascontiguousarray(v.reshape(sc.vertices.shape).astype(np.float32))
    sc._desc = None
    h = gpu_render(sc, 3, max_bounces=8, flags=abi.FLAG_HOST_BUILD)
    d = gpu_render(sc, 3, max_bounces=8, flags=abi.FLAG_GPU_BUILD)
    same = (h["beauty"].view(np.uint32) == d["beauty"].view(np.uint32)).all(-1)
    assert same.mean() >= 0.999, float(same.mean())


def test_device_builder_on_coincident_centroids_and_tiny_scenes():
    """Degenerate inputs of the split: every triangle's centroid in one place (no axis to bin on: the range is halved by position
    under the node's whole box), three triangles (one split), and duplicates of one triangle -- through the forced device build,
    against the host build."""
    base = scenes.soup(64, 64, 48, seed=5, hdri_size=(32, 16))
    v = base.vertices.reshape(-1, 3, 3).copy()
    for variant in ("same-centroid", "three", "duplicates"):
        sc = scenes.soup(64, 64, 48, seed=5, hdri_size=(32, 16))
        w = v.copy()
        if variant == "same-centroid":
            w = w - w.mean(1, keepdims=True) + np.array([0.0, 0.0, 3.0], np.float32)      # all centroids at (0, 0, 3), to rounding
            w[:, :, :] = np.round(w * 64) / 64                                             # ... exactly representable: centroids bit-equal for many
        elif variant == "duplicates":
            w[:] = w[0]
        if variant == "three":
            sc = scenes.soup(3, 64, 48, seed=5, hdri_size=(32, 16))
        else:
            sc.vertices = np.ascontiguousarray(w.reshape(sc.vertices.shape).astype(np.float32))
            sc._desc = None
        h = gpu_render(sc, 2, max_bounces=4, flags=abi.FLAG_HOST_BUILD)
        d = gpu_render(sc, 2, max_bounces=4, flags=abi.FLAG_GPU_BUILD)
        same = (h["beauty"].view(np.uint32) == d["beauty"].view(np.uint32)).all(-1)
        assert same.mean() >= 0.995, (variant, float(same.mean()))      # (coincident triangles: exact ties are resolved by tree order)
        assert (h["samples"] == d["samples"]).all()


def test_a_failed_default_device_build_falls_back_to_the_host_build(capfd):
    """The device builder is the default from 20 000 triangles up; if it does not deliver (simulated: er_debug_set_gpu_build_failure) the
    call still does -- the host builder takes over and the image is the same -- SILENTLY when the device was out of memory, with one
    line on stderr when the builder itself failed (a builder fault must not pass for an OOM); if the caller had forced the
    device build its failure is the call's: ER_ERR_HIP / ER_ERR_OOM with the reason."""
    lib = abi.load()
    sc = scenes.soup(30000, 120, 88, seed=8, hdri_size=(128, 64))
    ref = gpu_render(sc, 3, max_bounces=8)
    assert _accel(sc, 0)["builder"] == 1
    try:
        for kind, code, loud in ((1, abi.ER_ERR_HIP, True), (2, abi.ER_ERR_OOM, False)):
            lib.er_debug_set_gpu_build_failure(kind)
            capfd.readouterr()
            assert _accel(sc, 0)["builder"] == 0
            err = capfd.readouterr().err
            assert ("the device BVH build FAILED" in err) == loud, err
            got = gpu_render(sc, 3, max_bounces=8)
            same = (ref["beauty"].view(np.uint32) == got["beauty"].view(np.uint32)).all(-1)
            assert same.mean() >= 0.9995 and (ref["samples"] == got["samples"]).all()
            rm = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=abi.FLAG_GPU_BUILD))
            with pytest.raises(abi.ErError) as e:
                rm.start_rendering(sc)
            assert e.value.code == code and "simulated failure" in str(e.value)
            rm.close()
    finally:
        lib.er_debug_set_gpu_build_failure(0)
    assert _accel(sc, 0)["builder"] == 1

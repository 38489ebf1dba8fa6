"""How the super-tile edge of the streaming schedule's deal behaves on frames whose cost is NOT uniform (the BASELINE soup fills the
frame evenly; a super-tile goes whole to one XCD, so fewer, larger super-tiles sample the frame's cost more coarsely per XCD).
Scenes: the C2 soup seen from far away (it covers the middle of the frame, sky around it), from the side (half the frame), and the
C4 blobs; ms per sample pass at each edge.   python tools/supertile_nonuniform.py [edges...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elevenrender_amd import abi, render, scenes  # noqa: E402


def ms_per_pass(sc, max_bounces, spp=16):
    rm = render.RenderingManager(render.RenderParameters(max_bounces=max_bounces))
    rm.start_rendering(sc)
    rm.render(4)
    t0 = time.perf_counter()
    rm.render(spp)
    dt = time.perf_counter() - t0
    c = rm.counters()
    rm.close()
    return 1e3 * dt / spp, c["bounce_samples"]


def main():
    edges = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [8, 16, 0]      # 0 = the library's own choice (counted work of the first call)
    os.environ["ER_STREAM_VERBOSE"] = "1"      # the "[er_stream] counted work ..." line of the automatic choice goes to stderr
    cases = []
    cases.append(("C2 as benched (the soup fills the frame)", scenes.soup(1_000_000, 1920, 1080, seed=12345), 8))
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    sc.camera.position = abi.ErVec3(0.01, 0.02, -3.0)
    sc._desc = None
    cases.append(("soup from far away (middle of the frame)", sc, 8))
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    sc.camera.position = abi.ErVec3(1.2, 0.02, -1.5)
    sc._desc = None
    cases.append(("soup off to one side", sc, 8))
    cases.append(("C1 Cornell box at 1920x1080", scenes.cornell(1920, 1080), 5))
    if "--c4" in sys.argv:
        cases.append(("C4 (10 M triangles, 4K)", scenes.blob_instances(), 8))
    for name, sc, mb in cases:
        for e in edges:
            if e > 0:
                os.environ["ER_STREAM_SUPER_TILE"] = str(e)
            else:
                os.environ.pop("ER_STREAM_SUPER_TILE", None)      # 0 = the library's own choice
            ms, n = ms_per_pass(sc, mb)
            print(f"{name}: edge {e}: {ms:.3f} ms per pass, {n} bounce samples in all", flush=True)


if __name__ == "__main__":
    main()

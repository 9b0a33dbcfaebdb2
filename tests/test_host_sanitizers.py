"""The host-only parts of libmipt under AddressSanitizer + UndefinedBehaviorSanitizer (CPU): BVH builders, 4-wide collapse,
validation and refit on awkward inputs; the RGBE / OBJ / PLY parsers on valid, truncated and randomly damaged input
(tests/host_sanitize.cpp).  GPU code is not sanitised here (no GPU sanitizer on this pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host_sanitize.cpp"),
           os.path.join(ROOT, "cudapathtracer_amd", "csrc", "bvh_builder.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", MIPT_BUILD_THREADS="4"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "no sanitizer report" in r.stdout


def test_spatial_splits_cannot_multiply_coincident_triangles(ptlib):
    """a pile of identical triangles: spatial splits used to cut it again and again (300 triangles -> 56 100 references,
    29 451 nodes, seconds of build time); the duplication allowance (triangles + 64, handed down in proportion to the
    references on either side, so independent of the thread count) bounds it"""
    import numpy as np
    data = ptlib.builtin_scene(0)
    v, i, m = data.meshes[0]
    tri = v[:3].copy()
    verts = np.tile(tri, 300)
    idx = np.arange(900, dtype=np.uint32)
    pile = ptlib.SceneData([(verts, idx, m)], data.sky)
    for width in (0, 2):
        sc = ptlib.Scene(None, pile, node_width=width)
        nodes, refs = sc.bvh()
        assert 300 <= len(refs) <= 300 + 300 + 64 and len(nodes) < 2000
    # a regular scene is nowhere near its allowance
    sc = ptlib.Scene(None, ptlib.builtin_scene(1))
    assert len(sc.bvh()[1]) < 1.15 * ptlib.builtin_scene(1).triangle_count

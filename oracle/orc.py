"""ctypes binding of liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle.h).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
The scene structs have the same layout as libmipt's, so tests hand the SAME numpy arrays to both sides.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
lib = C.CDLL(LIB_PATH)

ISECT_REF, ISECT_FLAT, ISECT_BRUTE = 0, 1, 2
RNG_DFS, RNG_NODE = 0, 1
MATH_LIBM, MATH_DET = 0, 1


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("roughness", C.c_float), ("metallic", C.c_float), ("opacity", C.c_float),
                ("emissiveColor", C.c_float * 3), ("emissiveStrength", C.c_float), ("n", C.c_float),
                ("translucent", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class SkyLight(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("dir", C.c_float * 3), ("intensity", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fov", C.c_float)]


class Settings(C.Structure):
    _fields_ = [("cam", Camera), ("viewportWidth", C.c_uint32), ("viewportHeight", C.c_uint32),
                ("frameID", C.c_uint32), ("seed", C.c_uint32), ("maxDepth", C.c_uint32)]


class MeshDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertexCnt", C.c_uint32), ("indices", C.c_void_p),
                ("indicesCnt", C.c_uint32), ("material", Material)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("meshCnt", C.c_uint32), ("sky", SkyLight),
                ("envMap", C.c_void_p), ("envW", C.c_uint32), ("envH", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("anyhit_rays", C.c_uint64), ("emitfirst_rays", C.c_uint64)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<i4"), ("mesh", "<i4"), ("front", "<i4")])
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])

_vp = C.c_void_p
lib.orc_tea16.argtypes = [C.c_uint32, C.c_uint32]
lib.orc_tea16.restype = C.c_uint32
lib.orc_lcg_floats.argtypes = [C.c_uint32, C.c_uint32, _vp]
lib.orc_lcg_floats.restype = None
lib.orc_node_seed.argtypes = [C.c_uint32, C.c_uint32]
lib.orc_node_seed.restype = C.c_uint32
lib.orc_math_eval.argtypes = [C.c_int, C.c_int, _vp, _vp, C.c_uint32, _vp]
lib.orc_math_eval.restype = None
lib.orc_make_mesh.argtypes = [C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32)]
lib.orc_make_mesh.restype = C.c_uint32
lib.orc_builtin_scene_mesh_count.argtypes = [C.c_int]
lib.orc_builtin_scene_mesh_count.restype = C.c_uint32
lib.orc_builtin_mesh.argtypes = [C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(Material)]
lib.orc_builtin_mesh.restype = C.c_uint32
lib.orc_builtin_sky.argtypes = [C.c_int, C.POINTER(SkyLight)]
lib.orc_builtin_sky.restype = None
lib.orc_make_camera.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
lib.orc_make_camera.restype = None
lib.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
lib.orc_scene_create.restype = _vp
lib.orc_scene_destroy.argtypes = [_vp]
lib.orc_scene_destroy.restype = None
lib.orc_scene_triangle_count.argtypes = [_vp]
lib.orc_scene_triangle_count.restype = C.c_uint32
lib.orc_scene_set_flat_bvh.argtypes = [_vp, _vp, C.c_uint32, _vp, C.c_uint32]
lib.orc_check_flat_bvh.argtypes = [_vp, C.c_char_p, C.c_uint32]
lib.orc_scene_set_flat_width.argtypes = [_vp, C.c_uint32]
lib.orc_scene_set_estimator.argtypes = [_vp, C.c_int]
lib.orc_scene_set_anyhit_last_level.argtypes = [_vp, C.c_int]
lib.orc_scene_set_emitter_tree.argtypes = [_vp, C.c_uint32, C.c_uint32]
lib.orc_scene_get_env_cdf.argtypes = [_vp, _vp, C.c_uint64]
lib.orc_scene_get_env_cdf.restype = C.c_uint64
lib.orc_scene_get_emitters.argtypes = [_vp, _vp, _vp, C.c_uint64]
lib.orc_scene_get_emitters.restype = C.c_uint64
lib.orc_trace_rays.argtypes = [_vp, C.c_int, _vp, _vp, C.c_uint32, _vp, C.POINTER(Stats), C.c_int]
lib.orc_render.argtypes = [_vp, C.POINTER(Settings), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                           C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(Stats), C.c_int]


def _ptr(a):
    return None if a is None else a.ctypes.data


def tea16(a, b):
    return lib.orc_tea16(a, b)


def lcg_floats(state, n):
    out = np.zeros(n, np.float32)
    lib.orc_lcg_floats(state, n, _ptr(out))
    return out


def math_eval(fn, mode, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float32)
    out = np.zeros_like(a)
    lib.orc_math_eval(fn, mode, _ptr(a), _ptr(b), len(a), _ptr(out))
    return out


def builtin_meshes(which):
    """The oracle's own restatement of the reference scenes: list of (vertices, indices, Material)."""
    out = []
    for m in range(lib.orc_builtin_scene_mesh_count(which)):
        icnt = C.c_uint32(0)
        nv = lib.orc_builtin_mesh(which, m, None, 0, None, 0, C.byref(icnt), None)
        v = np.zeros(nv, VERTEX_DTYPE)
        i = np.zeros(icnt.value, np.uint32)
        mat = Material()
        lib.orc_builtin_mesh(which, m, _ptr(v), nv, _ptr(i), icnt.value, C.byref(icnt), C.byref(mat))
        out.append((v, i, mat))
    return out


def builtin_sky(which):
    s = SkyLight()
    lib.orc_builtin_sky(which, C.byref(s))
    return s


def make_camera(pos, azimuth, zenith, fov):
    cam = Camera()
    lib.orc_make_camera((C.c_float * 3)(*pos), azimuth, zenith, fov, C.byref(cam))
    return cam


def make_settings(cam, w, h, max_depth, frame=0):
    s = Settings()
    C.memmove(C.addressof(s.cam), C.addressof(cam), C.sizeof(Camera))
    s.viewportWidth, s.viewportHeight, s.frameID, s.seed, s.maxDepth = w, h, frame, 0, max_depth
    return s


class Scene:
    """Oracle scene built from the same arrays the product gets (cudapathtracer_amd.SceneData)."""

    def __init__(self, data):
        self.data = data
        arr = (MeshDesc * max(len(data.meshes), 1))()
        for k, (v, i, m) in enumerate(data.meshes):
            arr[k].vertices, arr[k].vertexCnt = v.ctypes.data, len(v)
            arr[k].indices, arr[k].indicesCnt = i.ctypes.data, len(i)
            C.memmove(C.addressof(arr[k].material), C.addressof(m), C.sizeof(Material))
        d = SceneDesc()
        d.meshes = C.cast(arr, C.POINTER(MeshDesc))
        d.meshCnt = len(data.meshes)
        C.memmove(C.addressof(d.sky), C.addressof(data.sky), C.sizeof(SkyLight))
        if data.env is not None:
            d.envMap = data.env.ctypes.data
            d.envH, d.envW = data.env.shape[0], data.env.shape[1]
        self.h = lib.orc_scene_create(C.byref(d))

    def set_flat_bvh(self, nodes, refs, width=2):
        """width: node layout of `nodes` (2 = BVH2, 4 = the product's quantised 4-wide nodes)."""
        nodes = np.ascontiguousarray(nodes, np.float32)
        refs = np.ascontiguousarray(refs, np.float32)
        lib.orc_scene_set_flat_bvh(self.h, _ptr(nodes), len(nodes), _ptr(refs), len(refs))
        if lib.orc_scene_set_flat_width(self.h, width) != 0:
            raise ValueError("width must be 2 or 4")

    def set_anyhit_last_level(self, on=True):
        """FLAT intersector: last-level rays end at their first hit when no mesh emits (the product's shortcut); counts only."""
        lib.orc_scene_set_anyhit_last_level(self.h, 1 if on else 0)

    def set_emitter_tree(self, root_ref, n_refs_main):
        """counting option: the product's "emitters first" rule for last-level rays (needs set_anyhit_last_level and flat
        arrays that hold the emitters' tree behind the main tree); root_ref 0 switches it off"""
        if lib.orc_scene_set_emitter_tree(self.h, int(root_ref), int(n_refs_main)) != 0:
            raise ValueError("emitters' tree does not lie inside the flat arrays")

    def set_estimator(self, estimator):
        """0 = the reference's estimator, 1 = light sampling, 2 = VNDF + cosine lobes, 3 = both (specification: include/mipt.h)."""
        if lib.orc_scene_set_estimator(self.h, estimator) != 0:
            raise ValueError("estimator must be 0, 1, 2 or 3")

    def env_cdf(self):
        n = lib.orc_scene_get_env_cdf(self.h, None, 0)
        out = np.zeros(n, np.float32)
        lib.orc_scene_get_env_cdf(self.h, _ptr(out), n)
        return out

    def emitters(self):
        n = lib.orc_scene_get_emitters(self.h, None, None, 0)
        tris, cdf = np.zeros((n, 15), np.float32), np.zeros(n + 1 if n else 0, np.float32)
        if n:
            lib.orc_scene_get_emitters(self.h, _ptr(tris), _ptr(cdf), n)
        return tris, cdf

    def check_flat_bvh(self):
        buf = C.create_string_buffer(256)
        rc = lib.orc_check_flat_bvh(self.h, buf, 256)
        return rc, buf.value.decode()

    def trace_rays(self, isect, origins, dirs, nthreads=8):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        out = np.zeros(len(o), HIT_DTYPE)
        st = Stats()
        lib.orc_trace_rays(self.h, isect, _ptr(o), _ptr(d), len(o), _ptr(out), C.byref(st), nthreads)
        return out, st

    def render(self, settings, isect, rng, math, first_frame=0, n_frames=1, rect=None, acc=None, want_ldr=True,
               nthreads=8):
        """Returns dict(acc, ldr, sample, rays, stats).  settings may be either binding's Settings (same layout)."""
        w, h = settings.viewportWidth, settings.viewportHeight
        s = Settings()
        C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(Settings))
        x0, y0, x1, y1 = rect if rect else (0, 0, w, h)
        acc = np.zeros((h, w, 4), np.float32) if acc is None else acc
        ldr = np.zeros((h, w, 4), np.uint8) if want_ldr else None
        sample = np.zeros((h, w, 3), np.float32)
        rays = np.zeros((h, w), np.uint32)
        st = Stats()
        lib.orc_render(self.h, C.byref(s), isect, rng, math, first_frame, n_frames, x0, y0, x1, y1,
                       _ptr(acc), _ptr(ldr), _ptr(sample), _ptr(rays), C.byref(st), nthreads)
        return dict(acc=acc, ldr=ldr, sample=sample, rays=rays, stats=st)

    def close(self):
        if self.h:
            lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

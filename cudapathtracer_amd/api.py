"""ctypes binding of libmipt.so (include/mipt.h).

Plumbing for tests and bench.py only: the product is the C-ABI library and the header-only C++ classes in
cudapathtracer_amd/cpp/ that re-create the reference's `CUDATracer` API on top of it.  There is no CPU fallback:
if libmipt.so is missing this module raises at import.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPT_LIB") or os.path.join(_HERE, "libmipt.so")     # MIPT_LIB: an instrumented build (make prof)
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C cudapathtracer_amd/csrc` (or __graft_entry__.build()); "
        "there is no CPU fallback for the path tracer")
lib = C.CDLL(LIB_PATH)
ABI_VERSION = 3         # include/mipt.h MIPT_ABI_VERSION this binding's struct layouts were written against
lib.mipt_abi_version.restype = C.c_uint32
if lib.mipt_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {lib.mipt_abi_version()}, this binding was written for {ABI_VERSION}: rebuild the library")


# ---- PODs (sizes asserted against the reference layout in tests/test_abi.py) ------------------------------------
class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("uv", C.c_float * 2)]


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
                ("envMap", C.c_void_p), ("envW", C.c_uint32), ("envH", C.c_uint32),
                ("bvhBuilder", C.c_uint32), ("maxLeafSize", C.c_uint32), ("splitAlpha", C.c_float), ("sahTriCost", C.c_float),
                ("nodeWidth", C.c_uint32), ("bvhCachePath", C.c_char_p)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("frames", C.c_uint64), ("kernel_ms", C.c_double), ("launches", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("kernel_span_ms", C.c_double)]


class IssueResult(C.Structure):
    _fields_ = [("ms", C.c_double), ("wave_instructions", C.c_uint64), ("waves", C.c_uint32), ("simds", C.c_uint32),
                ("median_cycles", C.c_double), ("max_cycles", C.c_double), ("median_realtime_ticks", C.c_double),
                ("clock_ghz", C.c_double)]


class LevelProfile(C.Structure):
    _fields_ = [("rays", C.c_uint64 * 32), ("ms", C.c_double * 32), ("launches", C.c_uint64 * 32), ("accumulate_ms", C.c_double)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<i4"), ("mesh", "<i4"), ("front", "<i4")])
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])

BVH_SBVH, BVH_MEDIAN, BVH_LBVH_GPU = 0, 1, 2
RNG_NODE, RNG_DFS = 0, 1

_vp = C.c_void_p
lib.mipt_create.argtypes = [C.c_int, C.POINTER(_vp)]
lib.mipt_destroy.argtypes = [_vp]
lib.mipt_destroy.restype = None
lib.mipt_last_error.argtypes = [_vp]
lib.mipt_last_error.restype = C.c_char_p
lib.mipt_set_stream.argtypes = [_vp, _vp]
lib.mipt_synchronize.argtypes = [_vp]
lib.mipt_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
lib.mipt_get_counters.argtypes = [_vp, C.POINTER(Counters)]
lib.mipt_reset_counters.argtypes = [_vp]
lib.mipt_get_level_profile.argtypes = [_vp, C.POINTER(LevelProfile)]
lib.mipt_scene_create.argtypes = [_vp, C.POINTER(SceneDesc), C.POINTER(_vp)]
lib.mipt_scene_destroy.argtypes = [_vp]
lib.mipt_scene_destroy.restype = None
lib.mipt_scene_refit.argtypes = [_vp, _vp]
lib.mipt_set_tile_partition.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32]
lib.mipt_scene_get_bvh.argtypes = [_vp, _vp, C.POINTER(C.c_uint32), _vp, C.POINTER(C.c_uint32)]
lib.mipt_scene_triangle_count.argtypes = [_vp]
lib.mipt_scene_triangle_count.restype = C.c_uint32
lib.mipt_scene_bvh_was_cached.argtypes = [_vp]
lib.mipt_scene_get_emitter_tree.argtypes = [_vp, _vp, C.POINTER(C.c_uint32), _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
lib.mipt_scene_get_env_cdf.argtypes = [_vp, _vp, C.c_uint64]
lib.mipt_scene_get_env_cdf.restype = C.c_uint64
lib.mipt_scene_get_emitters.argtypes = [_vp, _vp, _vp, C.c_uint64]
lib.mipt_scene_get_emitters.restype = C.c_uint64
lib.mipt_scene_bvh_width.argtypes = [_vp]
lib.mipt_scene_bvh_width.restype = C.c_uint32
lib.mipt_trace.argtypes = [_vp, _vp, C.POINTER(Settings), _vp, _vp]
lib.mipt_render.argtypes = [_vp, _vp, C.POINTER(Settings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vp, _vp]
lib.mipt_trace_static.argtypes = [_vp, _vp, C.POINTER(Settings), _vp, _vp]
lib.mipt_buffer_async_gpu_push_on.argtypes = [_vp, _vp]
lib.mipt_buffer_set_cpu_data.argtypes = [_vp, _vp]
lib.mipt_buffer_set_gpu_data.argtypes = [_vp, _vp]
lib.mipt_tonemap.argtypes = [_vp, _vp, C.c_float, C.c_uint32, _vp]
lib.mipt_last_sample.argtypes = [_vp, _vp, _vp, C.c_uint32]
lib.mipt_trace_rays.argtypes = [_vp, _vp, _vp, _vp, C.c_uint32, _vp]
lib.mipt_buffer_create.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
lib.mipt_buffer_destroy.argtypes = [_vp]
lib.mipt_buffer_destroy.restype = None
for _n in ("mipt_buffer_cpu_data", "mipt_buffer_gpu_data", "mipt_buffer_mutable_cpu_data", "mipt_buffer_mutable_gpu_data"):
    getattr(lib, _n).argtypes = [_vp]
    getattr(lib, _n).restype = _vp
lib.mipt_buffer_head.argtypes = [_vp]
lib.mipt_buffer_size.argtypes = [_vp]
lib.mipt_buffer_size.restype = C.c_size_t
lib.mipt_buffer_async_gpu_push.argtypes = [_vp]
lib.mipt_malloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
lib.mipt_free.argtypes = [_vp, _vp]
lib.mipt_upload.argtypes = [_vp, _vp, _vp, C.c_size_t]
lib.mipt_download.argtypes = [_vp, _vp, _vp, C.c_size_t]
lib.mipt_memset.argtypes = [_vp, _vp, C.c_int, C.c_size_t]
lib.mipt_make_mesh.argtypes = [C.c_int, _vp, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32)]
lib.mipt_make_mesh.restype = C.c_uint32
lib.mipt_builtin_mesh_count.argtypes = [C.c_int]
lib.mipt_builtin_mesh_count.restype = C.c_uint32
lib.mipt_builtin_mesh.argtypes = [C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(Material)]
lib.mipt_builtin_mesh.restype = C.c_uint32
lib.mipt_builtin_sky.argtypes = [C.c_int, C.POINTER(SkyLight)]
lib.mipt_builtin_sky.restype = None
lib.mipt_make_camera.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
lib.mipt_make_camera.restype = None
lib.mipt_make_synthetic_sky.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _vp]
lib.mipt_make_synthetic_sky.restype = None
lib.mipt_make_grid_mesh.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_uint32, _vp, _vp]
lib.mipt_make_grid_mesh.restype = C.c_uint32
lib.mipt_rgbe_decode.argtypes = [_vp, C.c_size_t, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
lib.mipt_rgbe_encode.argtypes = [_vp, C.c_uint32, C.c_uint32, _vp, C.c_size_t]
lib.mipt_rgbe_encode.restype = C.c_size_t
lib.mipt_test_math.argtypes = [_vp, C.c_int, _vp, _vp, C.c_uint32, _vp]
lib.mipt_test_rcp_sweep.argtypes = [_vp, C.c_int, C.POINTER(C.c_uint64)]
lib.mipt_test_stack_guard.argtypes = [_vp, C.POINTER(C.c_uint64)]
lib.mipt_test_issue.argtypes = [_vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(IssueResult)]


class MiptError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data


# ---- scene data as numpy, shared by the product and (through tests) the oracle ---------------------------------
class SceneData:
    """Plain arrays describing a scene: the same bytes are handed to libmipt and to the oracle."""

    def __init__(self, meshes, sky, env=None):
        self.meshes = meshes            # list of (vertices[VERTEX_DTYPE], indices[uint32], Material)
        self.sky = sky                  # SkyLight
        self.env = env                  # float32 [H, W, 3] or None

    @property
    def triangle_count(self):
        return sum(len(i) // 3 for _, i, _ in self.meshes)

    def fill_desc(self, desc_cls, mesh_cls, material_cls=None, sky_cls=None):
        """Build a ctypes scene desc of the given struct classes (mipt's or the oracle's, same layout).
        Returns (desc, keepalive)."""
        arr = (mesh_cls * max(len(self.meshes), 1))()
        for k, (v, i, m) in enumerate(self.meshes):
            arr[k].vertices = v.ctypes.data
            arr[k].vertexCnt = len(v)
            arr[k].indices = i.ctypes.data
            arr[k].indicesCnt = len(i)
            C.memmove(C.addressof(arr[k].material), C.addressof(m), C.sizeof(Material))
        d = desc_cls()
        d.meshes = C.cast(arr, type(d.meshes))
        d.meshCnt = len(self.meshes)
        C.memmove(C.addressof(d.sky), C.addressof(self.sky), C.sizeof(SkyLight))
        if self.env is not None:
            d.envMap = self.env.ctypes.data
            d.envH, d.envW = self.env.shape[0], self.env.shape[1]
        return d, arr


def builtin_scene(which, env=None):
    """0 = Cornell box (CornellBox/tracer.cpp:16-143), 1 = DefaultScene (DefaultScene/tracer.cpp:28-123)."""
    meshes = []
    for m in range(lib.mipt_builtin_mesh_count(which)):
        icnt = C.c_uint32(0)
        nv = lib.mipt_builtin_mesh(which, m, None, 0, None, 0, C.byref(icnt), None)
        v = np.zeros(nv, VERTEX_DTYPE)
        i = np.zeros(icnt.value, np.uint32)
        mat = Material()
        lib.mipt_builtin_mesh(which, m, _ptr(v), nv, _ptr(i), icnt.value, C.byref(icnt), C.byref(mat))
        meshes.append((v, i, mat))
    sky = SkyLight()
    lib.mipt_builtin_sky(which, C.byref(sky))
    return SceneData(meshes, sky, env)


def synthetic_sky(w=2048, h=1024, seed=0x5EED):
    img = np.zeros((h, w, 3), np.float32)
    lib.mipt_make_synthetic_sky(w, h, seed, _ptr(img))
    return img


def rgbe_decode(data):
    """bytes of a Radiance .hdr file -> float32 [H, W, 3] (top row first); raises MiptError on malformed input."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(0), C.c_uint32(0)
    if lib.mipt_rgbe_decode(_ptr(buf), len(buf), None, C.byref(w), C.byref(h)) != 0:
        raise MiptError("malformed RGBE data")
    img = np.zeros((h.value, w.value, 3), np.float32)
    lib.mipt_rgbe_decode(_ptr(buf), len(buf), _ptr(img), C.byref(w), C.byref(h))
    return img


def rgbe_encode(img):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    n = lib.mipt_rgbe_encode(_ptr(img), w, h, None, 0)
    out = np.zeros(n, np.uint8)
    lib.mipt_rgbe_encode(_ptr(img), w, h, _ptr(out), n)
    return out.tobytes()


def grid_mesh(n, extent=8.0, height=1.0, seed=1):
    nv = (n + 1) * (n + 1)
    v = np.zeros(nv, VERTEX_DTYPE)
    i = np.zeros(n * n * 6, np.uint32)
    lib.mipt_make_grid_mesh(n, extent, height, seed, _ptr(v), _ptr(i))
    return v, i


def make_material(color=(1, 1, 1), roughness=0.5, metallic=0.0, opacity=1.0, n=2.5, emissive=(0, 0, 0), strength=0.0):
    m = Material()
    m.color[:] = color
    m.roughness, m.metallic, m.opacity, m.n = roughness, metallic, opacity, n
    m.emissiveColor[:] = emissive
    m.emissiveStrength = strength
    return m


def make_camera(pos, azimuth, zenith, fov):
    cam = Camera()
    p = (C.c_float * 3)(*pos)
    lib.mipt_make_camera(p, azimuth, zenith, fov, C.byref(cam))
    return cam


def make_settings(cam, w, h, max_depth, frame=0):
    s = Settings()
    C.memmove(C.addressof(s.cam), C.addressof(cam), C.sizeof(Camera))
    s.viewportWidth, s.viewportHeight, s.frameID, s.seed, s.maxDepth = w, h, frame, 0, max_depth
    return s


CORNELL_CAMERA = dict(pos=(-6.0, 0.0, 0.0), azimuth=0.0, zenith=float(np.float32(np.pi / 2)), fov=float(np.float32(np.pi / 3)))
DEFAULT_CAMERA = dict(pos=(-1.0, 1.0, 0.0), azimuth=0.0, zenith=float(np.float32(np.pi / 2)), fov=float(np.float32(np.pi / 3)))


# ---- handles -----------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self, tracer, data, builder=BVH_SBVH, max_leaf=0, split_alpha=0.0, tri_cost=0.0, bvh_cache=None, node_width=0):
        """tracer may be None: host-only scene (BVH build + get_bvh), usable without a GPU.
        bvh_cache: path of a BVH cache file (loaded if it matches these triangles and parameters, else written)."""
        self.h = None
        self.tracer = tracer
        self.data = data
        desc, keep = data.fill_desc(SceneDesc, MeshDesc)
        desc.bvhBuilder, desc.maxLeafSize, desc.splitAlpha, desc.sahTriCost = builder, max_leaf, split_alpha, tri_cost
        desc.bvhCachePath = os.fsencode(bvh_cache) if bvh_cache else None
        desc.nodeWidth = node_width
        h = _vp()
        rc = lib.mipt_scene_create(tracer.h if tracer else None, C.byref(desc), C.byref(h))
        if rc != 0:
            raise MiptError(f"mipt_scene_create: {rc} {tracer.last_error() if tracer else ''}")
        self.h = h

    def refit(self, data):
        """Same topology, moved vertices / edited materials: re-fits the BVH boxes and overwrites the device buffers."""
        desc, keep = data.fill_desc(SceneDesc, MeshDesc)
        rc = lib.mipt_scene_refit(self.h, C.byref(desc))
        if rc != 0:
            raise MiptError(f"mipt_scene_refit: {rc} {self.tracer.last_error() if self.tracer else ''}")
        self.data = data

    @property
    def bvh_width(self):
        return int(lib.mipt_scene_bvh_width(self.h))

    @property
    def bvh_was_cached(self):
        return bool(lib.mipt_scene_bvh_was_cached(self.h))

    def bvh(self):
        nn, nr = C.c_uint32(0), C.c_uint32(0)
        lib.mipt_scene_get_bvh(self.h, None, C.byref(nn), None, C.byref(nr))
        nodes = np.zeros((nn.value, 16), np.float32)
        refs = np.zeros((nr.value, 12), np.float32)
        lib.mipt_scene_get_bvh(self.h, _ptr(nodes), C.byref(nn), _ptr(refs), C.byref(nr))
        return nodes, refs

    def env_cdf(self):
        n = lib.mipt_scene_get_env_cdf(self.h, None, 0)
        out = np.zeros(n, np.float32)
        lib.mipt_scene_get_env_cdf(self.h, _ptr(out), n)
        return out

    def emitters(self):
        n = lib.mipt_scene_get_emitters(self.h, None, None, 0)
        tris, cdf = np.zeros((n, 15), np.float32), np.zeros(n + 1 if n else 0, np.float32)
        if n:
            lib.mipt_scene_get_emitters(self.h, _ptr(tris), _ptr(cdf), n)
        return tris, cdf

    def flat(self):
        """(nodes, refs, width): what the oracle's set_flat_bvh takes."""
        nodes, refs = self.bvh()
        return nodes, refs, self.bvh_width

    def emitter_tree(self):
        """(nodes, refs, root_ref) of the emitters' tree as it lies behind the main tree on the device; root_ref 0: none"""
        nn, nr, root = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        lib.mipt_scene_get_emitter_tree(self.h, None, C.byref(nn), None, C.byref(nr), C.byref(root))
        nodes = np.zeros((nn.value, 16), np.float32)
        refs = np.zeros((nr.value, 12), np.float32)
        if root.value:
            lib.mipt_scene_get_emitter_tree(self.h, _ptr(nodes) if nn.value else None, C.byref(nn), _ptr(refs), C.byref(nr), C.byref(root))
        return nodes, refs, root.value

    def flat_with_emitters(self):
        """(nodes, refs, width, root_ref, n_refs_main): the device's arrays -- main tree, then the emitters' tree -- for the
        oracle's set_flat_bvh + set_emitter_tree"""
        nodes, refs = self.bvh()
        en, er, root = self.emitter_tree()
        if not root:
            return nodes, refs, self.bvh_width, 0, len(refs)
        return np.concatenate([nodes, en]), np.concatenate([refs, er]), self.bvh_width, root, len(refs)

    def close(self):
        if self.h:
            lib.mipt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Buffer:
    """mipt_buffer: the CUDABuffer work-alike (lazy host / device mirror with the reference's 4-state head)."""

    def __init__(self, tracer, nbytes):
        self.h = None
        h = _vp()
        tracer._chk(lib.mipt_buffer_create(tracer.h, nbytes, C.byref(h)), "buffer_create")
        self.h, self.nbytes = h, nbytes

    def mutable_gpu_data(self):
        return lib.mipt_buffer_mutable_gpu_data(self.h)

    def gpu_data(self):
        return lib.mipt_buffer_gpu_data(self.h)

    def cpu_data(self, dtype=np.uint8):
        """CUDABuffer::cpu_data(): blocking device-to-host copy when the head is at the GPU; a view of the pinned mirror."""
        p = lib.mipt_buffer_cpu_data(self.h)
        if not p:
            raise MiptError("mipt_buffer_cpu_data failed")
        return np.frombuffer((C.c_uint8 * self.nbytes).from_address(p), dtype=dtype)

    @property
    def head(self):
        return int(lib.mipt_buffer_head(self.h))

    def close(self):
        if self.h:
            lib.mipt_buffer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Tracer:
    def __init__(self, device=0):
        self.h = None
        h = _vp()
        rc = lib.mipt_create(device, C.byref(h))
        if rc != 0:
            raise MiptError(f"mipt_create({device}) failed: {rc} (no GPU, or libmipt cannot reach it)")
        self.h = h
        self.device = device

    def last_error(self):
        return lib.mipt_last_error(self.h).decode()

    def _chk(self, rc, what):
        if rc != 0:
            raise MiptError(f"{what}: {rc} {self.last_error()}")

    def set_option(self, key, value):
        self._chk(lib.mipt_set_option(self.h, key.encode(), int(value)), f"set_option {key}")

    def set_tile_partition(self, tile_rows, n_parts, part):
        """Render only the row tiles part, part + n_parts, ... (tiles of tile_rows rows); n_parts = 1: the whole frame."""
        self._chk(lib.mipt_set_tile_partition(self.h, tile_rows, n_parts, part), "set_tile_partition")

    def set_stream(self, stream_ptr):
        self._chk(lib.mipt_set_stream(self.h, stream_ptr), "set_stream")

    def synchronize(self):
        self._chk(lib.mipt_synchronize(self.h), "synchronize")

    def counters(self):
        c = Counters()
        self._chk(lib.mipt_get_counters(self.h, C.byref(c)), "get_counters")
        return c

    def level_profile(self):
        p = LevelProfile()
        self._chk(lib.mipt_get_level_profile(self.h, C.byref(p)), "get_level_profile")
        return p

    def reset_counters(self):
        self._chk(lib.mipt_reset_counters(self.h), "reset_counters")

    def malloc(self, nbytes):
        p = _vp()
        self._chk(lib.mipt_malloc(self.h, nbytes, C.byref(p)), "malloc")
        return p.value

    def free(self, p):
        self._chk(lib.mipt_free(self.h, p), "free")

    def upload(self, dev, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(lib.mipt_upload(self.h, dev, arr.ctypes.data, arr.nbytes), "upload")

    def download(self, dev, shape, dtype):
        out = np.zeros(shape, dtype)
        self._chk(lib.mipt_download(self.h, out.ctypes.data, dev, out.nbytes), "download")
        return out

    def memset(self, dev, value, nbytes):
        self._chk(lib.mipt_memset(self.h, dev, value, nbytes), "memset")

    def trace(self, scene, settings, acc_dev, ldr_dev):
        self._chk(lib.mipt_trace(self.h, scene.h, C.byref(settings), acc_dev, ldr_dev), "trace")

    def trace_static(self, scene, settings, acc_dev, ldr_dev):
        """mipt_trace_static: the allocation-free variant of trace (<= 65536 pixels, maxDepth <= 8)."""
        self._chk(lib.mipt_trace_static(self.h, scene.h, C.byref(settings), acc_dev, ldr_dev), "trace_static")

    def render(self, scene, settings, first_frame, n_frames, acc_dev, ldr_dev=None, frame_stride=1, sum_mode=0):
        self._chk(lib.mipt_render(self.h, scene.h, C.byref(settings), first_frame, n_frames, frame_stride, sum_mode,
                                  acc_dev, ldr_dev), "render")

    def tonemap(self, acc_dev, scale, n_pixels, ldr_dev):
        self._chk(lib.mipt_tonemap(self.h, acc_dev, scale, n_pixels, ldr_dev), "tonemap")

    def last_sample(self, n_pixels, rays=False):
        rgb = np.zeros((n_pixels, 3), np.float32)
        cnt = np.zeros(n_pixels, np.uint32) if rays else None
        self._chk(lib.mipt_last_sample(self.h, _ptr(rgb), _ptr(cnt), n_pixels), "last_sample")
        return (rgb, cnt) if rays else rgb

    def trace_rays(self, scene, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = len(o)
        do, dd, dh = self.malloc(max(o.nbytes, 4)), self.malloc(max(d.nbytes, 4)), self.malloc(max(n * HIT_DTYPE.itemsize, 4))
        try:
            self.upload(do, o)
            self.upload(dd, d)
            self._chk(lib.mipt_trace_rays(self.h, scene.h, do, dd, n, dh), "trace_rays")
            return self.download(dh, n, HIT_DTYPE)
        finally:
            self.free(do), self.free(dd), self.free(dh)

    def test_math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        out = np.zeros_like(a)
        self._chk(lib.mipt_test_math(self.h, fn, _ptr(a), _ptr(b), len(a), _ptr(out)), "test_math")
        return out

    def test_rcp_sweep(self, steps):
        """(compared, differing, first differing input bits, differing by > 1 ulp) of rcp_newton<steps> vs 1.0f / x;
        steps = 4: of sqrt_unit vs sqrtf"""
        out = (C.c_uint64 * 4)()
        self._chk(lib.mipt_test_rcp_sweep(self.h, steps, out), "test_rcp_sweep")
        return tuple(int(v) for v in out)

    def test_stack_guard(self):
        """(bytes of the wave-stack arrays beyond the waves' regions, bytes of those that were overwritten, shade batches cut
        short for lack of stack room since reset_counters); the first two need option wave_stack_entries"""
        out = (C.c_uint64 * 3)()
        self._chk(lib.mipt_test_stack_guard(self.h, out), "test_stack_guard")
        return int(out[0]), int(out[1]), int(out[2])

    def test_issue(self, blocks_per_cu, iters, packed=0):
        """VALU issue calibration (csrc/microbench.hip): returns an IssueResult."""
        r = IssueResult()
        self._chk(lib.mipt_test_issue(self.h, packed, blocks_per_cu, iters, C.byref(r)), "test_issue")
        return r

    def close(self):
        if self.h:
            lib.mipt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

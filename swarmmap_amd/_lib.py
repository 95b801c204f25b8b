"""Loader for libswarmorb.so (in-tree build; fails loudly when it is missing)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SwarmOrbError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libswarmorb.so")


def build_library(force=False):
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return library_path()


class SoKeypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


class SoExtractorConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device", C.c_int32)]


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise SwarmOrbError(
            "libswarmorb.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C swarmmap_amd/csrc`. There is no CPU fallback." % path)
    # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 + libhsa-runtime64.so.
    # If /opt/rocm's copy were loaded first, torch would later bring up a second HSA runtime and see no GPU.
    # Importing torch first makes the dynamic loader bind our DT_NEEDED libamdhip64.so.7 to torch's copy.
    if os.environ.get("SWARMORB_STANDALONE_HIP", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: use /opt/rocm's runtime
            pass
    lib = C.CDLL(path)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    lib.so_status_string.restype = C.c_char_p
    lib.so_status_string.argtypes = [C.c_int]
    lib.so_last_error.restype = C.c_char_p
    lib.so_device_count.restype = C.c_int
    lib.so_extractor_create.argtypes = [C.POINTER(SoExtractorConfig), C.POINTER(vp)]
    lib.so_extractor_destroy.argtypes = [vp]
    lib.so_extractor_destroy.restype = None
    lib.so_extractor_capacity.argtypes = [vp]
    lib.so_extractor_quadtree_on_device.argtypes = [vp]
    lib.so_extractor_run.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
    lib.so_extractor_run_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
    lib.so_extractor_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.so_extractor_submit_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.so_extractor_collect.argtypes = [vp, vp, vp, C.c_int, ip]
    lib.so_extractor_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.so_extractor_level_size.argtypes = [vp, C.c_int, ip, ip]
    lib.so_extractor_get_level.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.so_extractor_get_candidates.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, ip]
    lib.so_extractor_set_profiling.argtypes = [vp, C.c_int]
    lib.so_extractor_get_profile.argtypes = [vp, vp]
    _LIB = lib
    return lib


def check(status):
    if status != 0:
        lib = load_library()
        raise SwarmOrbError("%s: %s" % (lib.so_status_string(status).decode(), lib.so_last_error().decode()))


def device_count():
    return int(load_library().so_device_count())


def device_host_cpus(device, slot=-1):
    """cpulist next to a device as a set of CPU numbers: the whole NUMA node (slot < 0) or one last-level-cache group of it
    (so_device_host_cpus, include/swarmorb.h).  None when the host does not say."""
    lib = load_library()
    lib.so_device_host_cpus.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(1024)
    if lib.so_device_host_cpus(int(device), int(slot), buf, len(buf)) != 0:
        return None
    cpus = set()
    for piece in buf.value.decode().split(","):
        lo, _, hi = piece.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def pin_process_near_device(device, groups=1):
    """Every thread of this process (the caller, the HIP runtime's helpers, whatever they start later) onto `groups`
    last-level-cache groups of the NUMA node next to the device, starting with group number `device`.  Call after the first
    HIP call of the process, when the runtime's threads exist.  The threads of one agent talk to each other and to the
    GPU through pinned memory all the time: on a two-socket host with sixteen L3 domains the OS's placement costs ~5 % of
    bench.py's frame rate and most of its run-to-run spread.  SWARMORB_NO_PIN=1 (or a host that does not say) leaves
    placement alone.  Returns the CPU set, or None."""
    import os
    if os.environ.get("SWARMORB_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = set()
    base = int(os.environ.get("SWARMORB_PIN_SLOT_BASE", "0"))  # several PROCESSES on one GPU: each names its own first group
    for g in range(max(1, int(groups))):
        got = device_host_cpus(device, int(device) + base + g)
        if got:
            cpus |= got
    try:
        cpus &= os.sched_getaffinity(0)  # never ask for CPUs a cgroup / taskset took away from the process
        if not cpus:
            return None
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass  # a thread that ended meanwhile, or one the kernel does not let us move
    except OSError:
        return None
    return cpus


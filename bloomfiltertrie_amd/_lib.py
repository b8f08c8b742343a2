"""ctypes binding of the C-ABI in include/bft_gpu.h (csrc/libbft_gpu.so).

There is no fallback: if the HIP library is missing or fails to load, importing the product API raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libbft_gpu.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "bft_gpu.h")

_lib = None

# name -> (restype, argtypes); one entry per declaration in include/bft_gpu.h
_P = C.c_void_p
SIGNATURES = {
    "bft_gpu_last_error": (C.c_char_p, []),
    "bft_gpu_device_count": (C.c_int, []),
    "bft_gpu_version": (C.c_char_p, []),
    "bft_gpu_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "bft_gpu_create_seeded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "bft_gpu_free": (None, [_P]),
    "bft_gpu_cache_release": (C.c_uint64, []),
    "bft_gpu_add_genome": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint32)]),
    "bft_gpu_genome_name": (C.c_int, [_P, C.c_uint32, C.c_char_p, C.c_uint32]),
    "bft_gpu_insert_kmers": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32]),
    "bft_gpu_insert_kmers_dev": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32]),
    "bft_gpu_insert_kmers_dev_async": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, _P]),
    "bft_gpu_build": (C.c_int, [_P]),
    "bft_gpu_query_presence": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "bft_gpu_query_presence_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_query_colors": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bft_gpu_query_colors_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_query_color_rows": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_query_branching": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_query_branching_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P]),
    "bft_gpu_query_sequences": (C.c_int, [_P, C.c_char_p, _P, C.c_uint64, C.c_double, C.c_int, _P]),
    "bft_gpu_query_sequences_dev": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint64, C.c_double, C.c_int, _P, _P]),
    "bft_gpu_load_bft": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "bft_gpu_write_bft": (C.c_int, [_P, C.c_char_p]),
    "bft_gpu_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "bft_gpu_debug_get_array": (C.c_int, [_P, C.c_char_p, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bft_gpu_query_color_rows_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P]),
    "bft_gpu_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "bft_gpu_footprint": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "bft_gpu_kernel_time": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "bft_gpu_build_time": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int]),
    "bft_gpu_build_stages": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "bft_gpu_extract": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bft_gpu_colorset": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bft_gpu_query_rows": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P]),
    "bft_gpu_colorset_annot": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bft_gpu_image_size": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "bft_gpu_image_pack": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "bft_gpu_image_unpack": (C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(_P)]),
    "bft_gpu_group_shard": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bft_gpu_group_create": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "bft_gpu_group_free": (None, [_P]),
    "bft_gpu_group_size": (C.c_int, [_P]),
    "bft_gpu_group_query_presence": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "bft_gpu_group_query_color_rows": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_group_query_branching": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "bft_gpu_group_member_device": (C.c_int, [_P, C.c_int]),
    "bft_gpu_group_member_footprint": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    "bft_gpu_group_member_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    "bft_gpu_group_query_presence_dev": (C.c_int, [_P, _P, _P, _P, _P]),
    "bft_gpu_group_query_color_rows_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "bft_gpu_group_query_branching_dev": (C.c_int, [_P, _P, _P, _P, _P, _P]),
}


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "all"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; it must be the one this process initialises, otherwise torch later
    # reports "No HIP GPUs are available" next to an already-initialised system runtime.  torch is plumbing
    # (device memory, streams, torch.distributed), so load it first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("BFT_GPU_LIB") or LIB_PATH  # tuning experiments only: another build of the same sources
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C {CSRC}` (hipcc, gfx950). "
            "bloomfiltertrie_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class BFTError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().bft_gpu_last_error()
        raise BFTError(f"bft_gpu error {rc}: {msg.decode() if msg else ''}")

"""ur_allgather_rows / ur_allgather_rows_bytes (include/ur_hotpath.h) through a real RCCL communicator made with ctypes on
librccl: on the one-GPU test box the communicator has ONE rank (RCCL refuses two ranks on one device), which still runs
ncclAllGather on the context's stream, in place, and every argument check. The N-rank data path is the same call with
other (n_ranks, rank); its sharding arithmetic is covered by tests/test_dist_gloo.py on two gloo ranks."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    for name in ("/opt/rocm/lib/librccl.so", "librccl.so", "librccl.so.1"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)  # global: ur_allgather_rows resolves ncclAllGather with dlsym(RTLD_DEFAULT)
        except OSError:
            continue
    pytest.skip("librccl not found")


def test_allgather_rows_on_a_one_rank_communicator(hotpath):
    import torch
    from unclerenderer_amd import lib
    L = lib.load()
    rccl = _rccl()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        w, h = 256, 64
        rng = np.random.default_rng(3)
        hdr = torch.from_numpy(rng.integers(-2 ** 15, 2 ** 15, size=(h, w, 4), dtype=np.int16)).cuda()
        want = hdr.clone()
        assert L.ur_allgather_rows(hotpath.ctx, comm, C.c_void_p(hdr.data_ptr()), w, h, 1, 0) == lib.UR_OK, L.ur_last_error()
        torch.cuda.synchronize()
        assert torch.equal(hdr, want), "a one-rank in-place all-gather leaves the frame as it is"
        # the RGBA8 form: the tonemapped band (4 B/pixel)
        ldr = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31, size=(h, w), dtype=np.int32)).cuda()
        want8 = ldr.clone()
        assert L.ur_allgather_rows_bytes(hotpath.ctx, comm, C.c_void_p(ldr.data_ptr()), w * 4, h, 1, 0) == lib.UR_OK, L.ur_last_error()
        torch.cuda.synchronize()
        assert torch.equal(ldr, want8)
        # the direct form (grouped ncclSend / ncclRecv pairs, UR_GATHER_DIRECT): with one rank the group is empty, the symbols
        # are resolved and the group calls are made on the same communicator; the ring form through the _ex entry point
        UR_GATHER_RING, UR_GATHER_DIRECT = 0, 1
        assert L.ur_allgather_rows_bytes_ex(hotpath.ctx, comm, C.c_void_p(ldr.data_ptr()), w * 4, h, 1, 0, UR_GATHER_DIRECT) == lib.UR_OK, L.ur_last_error()
        assert L.ur_allgather_rows_bytes_ex(hotpath.ctx, comm, C.c_void_p(ldr.data_ptr()), w * 4, h, 1, 0, UR_GATHER_RING) == lib.UR_OK, L.ur_last_error()
        torch.cuda.synchronize()
        assert torch.equal(ldr, want8)
        assert L.ur_allgather_rows_bytes_ex(hotpath.ctx, comm, C.c_void_p(ldr.data_ptr()), w * 4, h, 1, 0, 2) == lib.UR_EINVAL  # no such mode
        assert L.ur_allgather_rows_bytes_ex(hotpath.ctx, comm, C.c_void_p(ldr.data_ptr()), w * 4, 65, 8, 0, UR_GATHER_DIRECT) == lib.UR_EINVAL
        # argument validation: every bad call is refused before RCCL is reached
        p = C.c_void_p(hdr.data_ptr())
        assert L.ur_allgather_rows(hotpath.ctx, None, p, w, h, 1, 0) == lib.UR_EINVAL          # no communicator
        assert L.ur_allgather_rows(hotpath.ctx, comm, None, w, h, 1, 0) == lib.UR_EINVAL       # no image
        assert L.ur_allgather_rows(hotpath.ctx, comm, p, w, h, 0, 0) == lib.UR_EINVAL          # zero ranks
        assert L.ur_allgather_rows(hotpath.ctx, comm, p, w, h, 8, 8) == lib.UR_EINVAL          # rank out of range
        assert L.ur_allgather_rows(hotpath.ctx, comm, p, w, 65, 8, 0) == lib.UR_EINVAL         # 8 does not divide 65 rows
        assert L.ur_allgather_rows(None, comm, p, w, h, 1, 0) == lib.UR_EINVAL                 # no context
        assert L.ur_allgather_rows_bytes(hotpath.ctx, comm, p, 0, h, 1, 0) == lib.UR_EINVAL
        assert b"bad argument" in L.ur_last_error()
        torch.cuda.synchronize()
        assert torch.equal(hdr, want)
    finally:
        rccl.ncclCommDestroy(comm)

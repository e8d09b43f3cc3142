"""One process per GPU: attach the cross-rank all-reduce to a :class:`DeviceMatrix`.

The sample axis N is sharded over ranks; every pass ends with ONE small all-reduce (K+1 doubles for
an evaluation pass, K(K+1)/2-ish blocks of 256 doubles + K for a Gram pass) done by RCCL on the
device buffers inside ``libmbar_hip.so`` (``ncclAllReduce`` over xGMI).  ``torch.distributed`` is used
only as the rendezvous that carries the 128-byte ``ncclUniqueId`` from rank 0 to the others (the
launcher contract is ``python -m torch.distributed.run``); if RCCL cannot be initialised the
all-reduce falls back to the host through the same process group, and says so.
"""
import logging
import os

import numpy as np

logger = logging.getLogger(__name__)


def shard_bounds(N_total, rank, nranks, align=16):
    """Column range [n0, n1) of ``rank``: contiguous, multiples of ``align`` except at the end."""
    per = -(-N_total // nranks)
    per = -(-per // align) * align
    n0 = min(N_total, rank * per)
    n1 = min(N_total, n0 + per)
    return n0, n1


def init_process_group_from_env(backend="gloo"):
    """Join the launcher's process group (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def attach_allreduce(dm, prefer="rccl"):
    """Give ``dm`` its cross-rank reduction.  Returns "none", "rccl" or "host"."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return "none"
    rank, nranks = dist.get_rank(), dist.get_world_size()
    if prefer == "rccl":
        ok = True
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            # single-node rendezvous: RCCL's bootstrap sockets may use the loopback interface (it is skipped by
            # default, and a network-less container has no other one); the data path is xGMI either way
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        payload = [None]
        if rank == 0:  # a failure here must still reach the broadcast below, or the other ranks wait forever
            try:
                from . import _lib
                import ctypes as C

                buf = C.create_string_buffer(128)
                _lib.check(_lib.load_library().mbar_comm_unique_id(buf))
                payload = [bytes(buf.raw)]
            except Exception as exc:  # pragma: no cover - needs RCCL
                logger.warning("RCCL unique id could not be created (%s); using the host all-reduce", exc)
        dist.broadcast_object_list(payload, src=0)
        if payload[0] is None:
            ok = False
        else:
            try:
                dm.comm_init_rccl(payload[0], rank, nranks)
            except Exception as exc:  # pragma: no cover - needs several GPUs
                logger.warning("RCCL initialisation failed on rank %d (%s); using the host all-reduce", rank, exc)
                ok = False
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return "rccl"

    def host_allreduce(arr, op):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)

    dm.set_host_allreduce(host_allreduce, rank, nranks)
    return "host"

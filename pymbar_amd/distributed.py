"""One process per GPU: rendezvous of the ranks and the cross-rank all-reduce of a :class:`DeviceMatrix`.

The sample axis N is sharded over ranks; every pass ends with ONE small all-reduce (K+1 doubles for an
evaluation pass, K(K+1)/2-ish blocks of 256 doubles for a Gram pass) done by RCCL on the device buffers
inside ``libmbar_hip.so`` (``ncclAllReduce`` over xGMI, on the compute stream).  What the ranks need from
the host side is tiny: the 128-byte ``ncclUniqueId`` has to travel from rank 0 to the others, and the
launcher contract wants a barrier and a max-over-ranks of the elapsed time.  :class:`HostGroup` does that
with the standard library only (TCP sockets; rank 0 is the hub of a star) from the launcher's environment
(``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT``, as set by the driver's elastic launcher or any other
one) -- this package imports no ML framework, not even for the rendezvous.

If RCCL cannot be initialised on every rank, ALL ranks drop their communicator and the same reduction runs on
the host through the :class:`HostGroup` (``"host"``); callers that must not fall back (``bench.py``) check the
returned kind.
"""
import hashlib
import hmac
import logging
import os
import socket
import struct
import time

import numpy as np

logger = logging.getLogger(__name__)

_MAGIC = b"MBARRDZV1"
_PORT_SPAN = 24  # ports tried above the base port


def shard_bounds(N_total, rank, nranks, align=16):
    """Column range [n0, n1) of ``rank``: contiguous, multiples of ``align`` except at the end."""
    per = -(-N_total // nranks)
    per = -(-per // align) * align
    n0 = min(N_total, rank * per)
    n1 = min(N_total, n0 + per)
    return n0, n1


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf.extend(chunk)
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


_MAX_MSG = 64 << 20  # the largest legitimate message is a Gram all-reduce of a few MB


def _recv_msg(sock):
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    if n < 0 or n > _MAX_MSG:  # a peer that is not one of ours (or a desynchronised stream): never allocate on its say-so
        raise ConnectionError(f"rendezvous: implausible message length {n}")
    return _recv_exact(sock, n) if n else b""


def _bind_candidates(addr):
    """Addresses rank 0 tries to listen on, in order.  ``MBAR_RDZV_BIND`` overrides everything.  A loopback ``MASTER_ADDR``
    binds loopback only (single node: nothing off the host can reach the hub).  Otherwise the interface ``MASTER_ADDR``
    resolves to -- unless that is a loopback alias (Debian's ``127.0.1.1`` entry for the own hostname) or not a local interface
    at all (a NAT / VIP / service address: ``bind`` fails), in which cases remote ranks could never connect to it and the
    wildcard address is used; the token handshake (launch parameters + ``MBAR_RDZV_SECRET``) authenticates peers either way."""
    override = os.environ.get("MBAR_RDZV_BIND")
    if override:
        return [override]
    if addr in ("localhost", "::1") or addr.startswith("127."):
        return ["127.0.0.1"]
    try:
        resolved = socket.gethostbyname(addr)
    except OSError:
        return ["0.0.0.0"]
    if resolved.startswith("127."):  # the own hostname mapped to a loopback alias: remote ranks need a real interface
        return ["0.0.0.0"]
    return [resolved, "0.0.0.0"]


class HostGroup:
    """Minimal process group over TCP (standard library only): rank 0 listens, the others connect.

    Collectives are star-shaped and reduce in rank order on rank 0, so every rank receives bit-identical
    results.  Meant for the rendezvous traffic of a single node (a few hundred bytes per call) and as the
    fallback transport of the per-pass all-reduce (a few KB); the data path proper is RCCL.

    Port: the launcher's ``MASTER_PORT`` usually belongs to the launcher's own store (the elastic launcher keeps it
    bound), so rank 0 binds the first free port in ``[base, base + 24)`` with ``base = MASTER_PORT + 1``
    (or ``MBAR_RDZV_PORT``) and the clients probe the same range; a handshake token derived from the launch
    (address, port, run id, world size -- plus ``MBAR_RDZV_SECRET`` when the launcher exports one, which ``bench.py``'s own
    spawner does with a random value) tells this group's hub from anything else listening there.  Rank 0 binds the interface
    of ``MASTER_ADDR`` (loopback for a single node; the wildcard address only when that name is not a usable local interface --
    see ``_bind_candidates`` -- or ``MBAR_RDZV_BIND`` says so).  ``timeout`` bounds the rendezvous; once the group stands, a
    collective may wait ``data_timeout`` for a slow peer (default ``MBAR_RDZV_DATA_TIMEOUT`` or 3600 s: ranks skew by minutes
    when one of them uploads tens of GB, runs a BAR initialisation or a CPU baseline first).  The wildcard address is a
    fallback for ONE situation -- the resolved address is not an interface of this host (``EADDRNOTAVAIL``) -- and is logged
    as a warning when no ``MBAR_RDZV_SECRET`` strengthens the handshake token.  Messages above 64 MB travel
    in pieces (a K x K all-reduce on the host transport is 8 K^2 bytes)."""

    def __init__(self, rank, world, addr="127.0.0.1", base_port=29501, token="", timeout=120.0, data_timeout=None):
        self.rank, self.world = int(rank), int(world)
        self._socks = {}     # rank 0: peer rank -> socket
        self._sock = None    # other ranks: socket to rank 0
        self._listener = None
        if self.world <= 1:
            return
        if data_timeout is None:
            data_timeout = float(os.environ.get("MBAR_RDZV_DATA_TIMEOUT", "3600"))
        tok = hashlib.sha256(f"{token}|{self.world}".encode()).digest()[:16]
        deadline = time.time() + timeout
        if self.rank == 0:
            import errno

            last = None
            candidates = _bind_candidates(addr)
            for i, bind_addr in enumerate(candidates):
                not_local = False
                for port in range(base_port, base_port + _PORT_SPAN):
                    try:
                        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                        ls.bind((bind_addr, port))
                        ls.listen(self.world)
                        self._listener = ls
                        break
                    except OSError as exc:
                        last = exc
                        ls.close()
                        if exc.errno == errno.EADDRNOTAVAIL:  # not an interface of this host: no port will do
                            not_local = True
                            break
                if self._listener is not None:
                    if bind_addr == "0.0.0.0" and not os.environ.get("MBAR_RDZV_SECRET") and not os.environ.get("MBAR_RDZV_BIND"):
                        import logging

                        logging.getLogger(__name__).warning(
                            "rendezvous hub listens on the wildcard address (MASTER_ADDR=%s is not a local interface) and no "
                            "MBAR_RDZV_SECRET is set: the handshake token derives from launch parameters only", addr)
                    break
                # the next candidate is the wildcard address: only because this one is not a local interface -- never because its
                # ports happen to be taken (EADDRINUSE on all of them is an error, not a reason to listen everywhere)
                if i + 1 < len(candidates) and not not_local:
                    break
            if self._listener is None:
                raise RuntimeError(f"rendezvous: no free port in [{base_port}, {base_port + _PORT_SPAN}): {last}")
            self._listener.settimeout(1.0)
            while len(self._socks) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError(f"rendezvous: {self.world - 1 - len(self._socks)} rank(s) never connected")
                try:
                    conn, _ = self._listener.accept()
                except socket.timeout:
                    continue
                try:
                    conn.settimeout(5.0)
                    hello = _recv_exact(conn, len(_MAGIC) + 16 + 4)
                    peer = struct.unpack("<i", hello[-4:])[0]
                    if hello[: len(_MAGIC)] != _MAGIC or not hmac.compare_digest(hello[len(_MAGIC):-4], tok) \
                            or not (0 < peer < self.world) or peer in self._socks:
                        conn.close()
                        continue
                    conn.sendall(b"OK")
                    conn.settimeout(data_timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._socks[peer] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()
        else:
            hello = _MAGIC + tok + struct.pack("<i", self.rank)
            host = "127.0.0.1" if addr == "localhost" else addr
            while self._sock is None:
                for port in range(base_port, base_port + _PORT_SPAN):
                    try:
                        s = socket.create_connection((host, port), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        s.settimeout(3.0)
                        s.sendall(hello)
                        if _recv_exact(s, 2) == b"OK":
                            s.settimeout(data_timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._sock = s
                            break
                    except (OSError, ConnectionError):
                        pass
                    s.close()
                if self._sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank 0 could not be reached")
                    time.sleep(0.05)

    # ---- construction from the launcher's environment ------------------------------------------------
    @classmethod
    def from_env(cls, timeout=120.0):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        mport = int(os.environ.get("MASTER_PORT", "29500"))
        base = int(os.environ.get("MBAR_RDZV_PORT", mport + 1))
        token = f"{addr}:{mport}:{os.environ.get('TORCHELASTIC_RUN_ID', '')}:{os.environ.get('MBAR_RDZV_SECRET', '')}"
        return cls(rank, world, addr=addr, base_port=base, token=token, timeout=timeout)

    # ---- collectives -----------------------------------------------------------------------------------
    def broadcast_bytes(self, payload, src=0):
        """``payload`` of rank ``src`` (bytes, or None meaning "nothing") on every rank."""
        if self.world <= 1:
            return payload
        if src != 0:  # route through the hub
            if self.rank == src:
                _send_msg(self._sock, b"\x01" + payload if payload is not None else b"\x00")
            if self.rank == 0:
                m = _recv_msg(self._socks[src])
                payload = m[1:] if m[:1] == b"\x01" else None
        if self.rank == 0:
            msg = b"\x01" + payload if payload is not None else b"\x00"
            for r in sorted(self._socks):
                _send_msg(self._socks[r], msg)
            return payload
        m = _recv_msg(self._sock)
        return m[1:] if m[:1] == b"\x01" else None

    def allreduce(self, arr, op="sum"):
        """In-place all-reduce of a float64 numpy array (``op`` "sum", "max" or "min"); rank order on the hub.  Arrays beyond
        the message bound travel in pieces (every rank cuts at the same places)."""
        if self.world <= 1:
            return arr
        if op not in ("sum", "max", "min"):
            raise ValueError(op)
        a = np.ascontiguousarray(arr, dtype=np.float64)
        flat = a.reshape(-1)
        res = np.empty_like(flat)
        piece = _MAX_MSG // 16  # doubles per message (half the bound)
        for lo in range(0, max(1, flat.size), piece):
            part = flat[lo:lo + piece]
            if self.rank == 0:
                acc = part.copy()
                for r in sorted(self._socks):
                    other = np.frombuffer(_recv_msg(self._socks[r]), dtype=np.float64)
                    if op == "sum":
                        acc += other
                    elif op == "max":
                        np.fmax(acc, other, out=acc)   # (like ncclMax: the non-NaN operand wins)
                    else:
                        np.fmin(acc, other, out=acc)
                out = acc.tobytes()
                for r in sorted(self._socks):
                    _send_msg(self._socks[r], out)
                res[lo:lo + piece] = acc
            else:
                _send_msg(self._sock, part.tobytes())
                res[lo:lo + piece] = np.frombuffer(_recv_msg(self._sock), dtype=np.float64)
        arr[...] = res.reshape(a.shape)
        return arr

    def barrier(self):
        if self.world > 1:
            self.allreduce(np.zeros(1), "sum")

    def close(self):
        for s in list(self._socks.values()) + [self._sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:  # pragma: no cover
                    pass
        self._socks, self._sock, self._listener = {}, None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _StdoutToStderr:
    """RCCL prints a version banner on the process's STDOUT when its first communicator is initialised; a launcher that parses
    rank 0's standard output (bench.py prints ONE JSON line there) must not find it: file descriptor 1 points at stderr meanwhile."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys

        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def attach_allreduce(dm, group, prefer="rccl"):
    """Give ``dm`` its cross-rank reduction over the ranks of ``group``.  Returns "none", "rccl" or "host".

    ``group`` is a :class:`HostGroup` (or any object with ``rank``, ``world``, ``broadcast_bytes`` and
    ``allreduce(array, op)``).  Collective: every rank of the group must call it.  When RCCL was preferred and "host" comes
    back, ``dm.rccl_error`` holds, on EVERY rank, the message of the lowest rank that failed (a launcher can print it)."""
    if group is None or group.world <= 1:
        return "none"
    dm.rccl_error = None
    rank, nranks = group.rank, group.world
    my_error = None
    if prefer == "rccl":
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            # single-node rendezvous: RCCL's bootstrap sockets may use the loopback interface (it is skipped by
            # default, and a network-less container has no other one); the data path is xGMI either way
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        payload = None
        if rank == 0:  # a failure here must still reach the broadcast below, or the other ranks wait forever
            try:
                import ctypes as C

                from . import _lib

                buf = C.create_string_buffer(128)
                with _StdoutToStderr():
                    _lib.check(_lib.load_library().mbar_comm_unique_id(buf))
                payload = bytes(buf.raw)
            except Exception as exc:  # pragma: no cover - needs RCCL
                logger.warning("RCCL unique id could not be created (%s); using the host all-reduce", exc)
                my_error = f"rank 0: ncclGetUniqueId: {exc}"
        payload = group.broadcast_bytes(payload, src=0)
        ok = payload is not None
        if ok:
            try:
                with _StdoutToStderr():
                    dm.comm_init_rccl(payload, rank, nranks)
            except Exception as exc:  # pragma: no cover - needs several GPUs
                logger.warning("RCCL initialisation failed on rank %d (%s); using the host all-reduce", rank, exc)
                my_error = f"rank {rank}: ncclCommInitRank: {exc}"
                ok = False
        elif my_error is None:
            my_error = f"rank {rank}: no unique id arrived from rank 0"
        flag = np.array([1.0 if ok else 0.0])
        group.allreduce(flag, "min")
        if flag[0] == 1.0:
            return "rccl"
        # the message of the lowest failing rank, on every rank
        who = np.array([float(rank) if not ok else float(nranks)])
        group.allreduce(who, "min")
        src = int(who[0]) if who[0] < nranks else 0
        msg = group.broadcast_bytes((my_error or "unknown").encode() if rank == src else None, src=src)
        dm.rccl_error = msg.decode(errors="replace") if msg else "unknown"
        # not every rank has a communicator: NOBODY may keep one (a rank that still issued ncclAllReduce while the
        # others reduce on the host would deadlock every later sweep)
        dm.comm_destroy()

    dm.set_host_allreduce(group.allreduce, rank, nranks)
    return "host"

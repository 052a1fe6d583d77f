"""Multi-GPU sharding of the stepper: one process per GPU, contiguous env ranges, the observation rows of all shards on every
rank after each control step (BASELINE.json configs[3]; the reference itself has no multi-GPU path).

Environments are fully independent (no cross-env term in quadrotor_multi.py), so rank r owns the global envs
[r*E, (r+1)*E) and steps them with `env_id_offset=r*E`; the counter-based RNG is keyed by the GLOBAL env id, so the
union of the shards is bit-identical to one un-sharded run (tests/test_hip_parity.py::test_determinism_and_sharding_invariance).
xGMI is point-to-point (7 links / GPU): the only traffic is each rank's observation shard going to its peers, no reduction.

Three transports, same result (`ObsExchange(transport=...)`):

  "fused" the step kernel itself stores its rows into every rank's receive window, straight from the workgroups' LDS stage, and the
          last workgroup raises the sequence flags (qs_set_obs_exchange): no launch besides the step.  Team kernels (batches up to
          ~8 waves per CU: the BASELINE shards); measured at world size 1: + 0.x us per step (profiles/r03*).

  "peer"  (default) the library's own exchange (include/quadswarm_exchange.h, csrc/qs_exchange.hip): every rank's receive window is
          mapped into its peers (hipIpc); ONE kernel per rank and step stores the rank's rows into all windows over the
          point-to-point links and raises per-source sequence flags.  No collective launch, no host round trip.
  "rccl"  torch.distributed.all_gather_into_tensor (backend "nccl" = RCCL) of the packed rows.

Wire format "bf16" (default; the fused policy encoder rounds its input to bf16 anyway) or "f32" (bit-exact rows).  At C4
(16384 drones x 54 columns per GPU) a step sends 1.77 MB per link in bf16, 3.54 MB in float32.

Either way the stepper writes the rows of consecutive steps alternately into two float32 staging buffers (qs_set_obs_target), so
that exchange(t) runs on a second stream under step(t+1); `capture()` records a whole segment [step -> exchange] x T into ONE HIP
graph (fork / join between the two streams inside the graph), which removes the per-step host cost of the launches.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import native

EXPORT_BYTES = 144   # QS_XCHG_EXPORT_BYTES
WIRE = {"f32": 0, "bf16": 1, "q8": 2}


def shard_range(total_envs, world_size, rank):
    """Contiguous env range [lo, hi) of `rank`; a whole env never straddles GPUs."""
    if total_envs % world_size:
        raise ValueError("total_envs must be divisible by the number of ranks")
    per = total_envs // world_size
    return rank * per, (rank + 1) * per


def _xcheck(rc):
    if rc != 0:
        raise native.QsError(f"exchange error {rc}: {native.lib().qs_xchg_last_error().decode()}")


def _dev_tensor(ptr, shape, wire, device):
    """zero-copy torch view of library memory: float32, bfloat16 (exposed as int16 through the CUDA array interface, then viewed) or the
    raw bytes of QS_WIRE_Q8 rows (uint8 [rows, row_bytes])"""
    if wire == "bf16":
        return torch.as_tensor(native._DevArray(ptr, shape, "<i2"), device=device).view(torch.bfloat16)
    if wire == "q8":
        return torch.as_tensor(native._DevArray(ptr, shape, "|u1"), device=device)
    return torch.as_tensor(native._DevArray(ptr, shape, "<f4"), device=device)


def wire_dtype(wire):
    return {"f32": torch.float32, "bf16": torch.bfloat16, "q8": torch.uint8}[wire]


def wire_row_bytes(cols, wire, q8=None):
    n = native.lib().qs_wire_row_bytes(cols, WIRE[wire], C.byref(q8) if q8 is not None else None)
    if n < 0:
        raise native.QsError("bad wire layout")
    return int(n)


def pack_rows(src, dst, stream=None, q8=None):
    """float32 rows [R, D] -> dst: float32 copy, bfloat16 round-to-nearest-even, or (dst uint8 [R, row_bytes], q8 = the layout) QS_WIRE_Q8
    rows, with the library's converter (qs_obs_pack_rows)."""
    wire = "bf16" if dst.dtype == torch.bfloat16 else ("q8" if dst.dtype == torch.uint8 else "f32")
    s = stream if stream is not None else torch.cuda.current_stream(src.device)
    rows, cols = (src.shape[0], src.shape[1]) if src.dim() == 2 else (src.numel(), 1)   # (f32 / bf16 are element-wise: any shape)
    _xcheck(native.lib().qs_obs_pack_rows(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), rows, cols, WIRE[wire],
                                          C.byref(q8) if q8 is not None else None, C.c_void_p(s.cuda_stream)))
    return dst


def unpack_rows(src, cols, wire, q8=None, out=None, stream=None):
    """wire rows -> float32 [R, cols] (qs_obs_unpack_rows): what a float32 consumer of gathered bf16 / q8 rows calls"""
    rows = src.shape[0]
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    s = stream if stream is not None else torch.cuda.current_stream(src.device)
    _xcheck(native.lib().qs_obs_unpack_rows(C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), rows, cols, WIRE[wire],
                                            C.byref(q8) if q8 is not None else None, C.c_void_p(s.cuda_stream)))
    return out


def quantize_rows_reference(rows, wire, q8=None):
    """The wire form of float32 rows computed with plain torch ops (any device) - the specification the library's converters, the push
    kernel and the fused epilogue are tested against.  bf16: torch's round-to-nearest-even; q8: see include/quadswarm_exchange.h."""
    rows = rows.float()
    if wire == "f32":
        return rows.clone()
    if wire == "bf16":
        return rows.to(torch.bfloat16)
    R, D = rows.shape
    q0, q1 = int(q8.q0), int(q8.q1)
    nq, c16 = q1 - q0, D - (q1 - q0)
    w16 = (c16 + 1) // 2
    out = torch.zeros((R, 4 * (w16 + (nq + 3) // 4)), dtype=torch.uint8, device=rows.device)
    other = torch.cat([rows[:, :q0], rows[:, q1:]], dim=1).to(torch.bfloat16).contiguous()
    out[:, :2 * c16] = other.view(torch.uint8).reshape(R, 2 * c16)
    if nq:
        scale = torch.tensor([float(np.float32(127.0 / float(q8.clip[a % 6]))) for a in range(nq)], dtype=torch.float32, device=rows.device)
        q = torch.clamp(torch.round(rows[:, q0:q1] * scale), -127.0, 127.0).to(torch.int8)
        out[:, 4 * w16:4 * w16 + nq] = q.view(torch.uint8)
    return out


def gather_window_handles(blob, world, group=None):
    """Every rank contributes the export blob of its endpoint - or None when it could not create / export one - and gets the blobs of all
    ranks in rank order.  A rank that failed still takes part in the collective (raising before it would leave the others waiting), and
    if ANY blob is missing every rank raises the same QsError, so that all of them take the fallback together."""
    blobs = [None] * world
    dist.all_gather_object(blobs, blob, group=group)
    missing = [r for r, b in enumerate(blobs) if b is None]
    if missing:
        raise native.QsError(f"rank(s) {missing} could not create / export their exchange windows")
    return blobs


class PeerExchange:
    """One endpoint of the peer-store exchange (thin wrapper of the qs_xchg_* C ABI)."""

    def __init__(self, rows, cols, world, rank, device=0, wire="bf16", q8=None, fenced=None):
        """q8: the native.WireQ8 layout (wire="q8" only; native.wire_q8_layout(cfg, obs_dim)).  fenced: True / False = the fenced / relaxed
        variant of the flag protocol (include/quadswarm_exchange.h, QS_XCHG_FENCED), None = whatever the environment says"""
        self.rows, self.cols, self.world, self.rank, self.device, self.wire, self.q8 = rows, cols, world, rank, device, wire, q8
        self._x = C.c_void_p()
        self._create(device, world, rank, rows, cols, wire, q8)
        if fenced is not None:   # explicit, through the C ABI (round 5 toggled the process-wide environment variable around the create call)
            _xcheck(native.lib().qs_xchg_set_fenced(self._x, 1 if fenced else 0))
        self.fenced = bool(native.lib().qs_xchg_get_fenced(self._x))   # the library's own value, not a second parse of QS_XCHG_FENCED
        self.row_bytes = wire_row_bytes(cols, wire, q8)
        self._views = {}

    def _create(self, device, world, rank, rows, cols, wire, q8):
        if wire == "q8":
            if q8 is None:
                raise ValueError("wire='q8' needs the row layout (native.wire_q8_layout)")
            _xcheck(native.lib().qs_xchg_create_q8(device, world, rank, rows, cols, C.byref(q8), C.byref(self._x)))
        else:
            _xcheck(native.lib().qs_xchg_create(device, world, rank, rows, cols, WIRE[wire], C.byref(self._x)))

    def close(self):
        if self._x:
            native.lib().qs_xchg_destroy(self._x)
            self._x = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- wiring ----
    def export(self):
        buf = C.create_string_buffer(EXPORT_BYTES)
        _xcheck(native.lib().qs_xchg_export(self._x, buf))
        return buf.raw

    def attach(self, blobs):
        """blobs: the export() of every rank, in rank order (other processes)."""
        assert len(blobs) == self.world and all(len(b) == EXPORT_BYTES for b in blobs)
        _xcheck(native.lib().qs_xchg_attach(self._x, C.create_string_buffer(b"".join(blobs), EXPORT_BYTES * self.world)))

    def attach_local(self, peer):
        """peer: a PeerExchange of the same process (its rank is taken from the object)."""
        _xcheck(native.lib().qs_xchg_attach_local(self._x, peer.rank, peer._x))

    def connect(self, group=None):
        """exchange the window handles over torch.distributed (any backend) and map the peers' windows"""
        blobs = [None] * self.world
        dist.all_gather_object(blobs, self.export(), group=group)
        self.attach(blobs)

    # ---- buffers ----
    def staging_ptr(self, slot):
        return native.lib().qs_xchg_staging(self._x, slot)

    def staging(self, slot):
        key = ("s", slot & 1)
        if key not in self._views:
            self._views[key] = _dev_tensor(self.staging_ptr(slot), (self.rows, self.cols), "f32", f"cuda:{self.device}")
        return self._views[key]

    def gathered(self, slot):
        """[world * rows, cols] rows of all ranks in rank order, wire dtype (q8: uint8 [world * rows, row_bytes], see unpack_rows);
        slot = seq & 1 of the wait() that returned them"""
        key = ("g", slot & 1)
        if key not in self._views:
            shape = (self.world * self.rows, self.row_bytes if self.wire == "q8" else self.cols)
            self._views[key] = _dev_tensor(native.lib().qs_xchg_gathered(self._x, slot), shape, self.wire, f"cuda:{self.device}")
        return self._views[key]

    # ---- protocol (all asynchronous on `stream`) ----
    @staticmethod
    def _s(stream):
        return C.c_void_p(stream.cuda_stream) if stream is not None else None

    def push(self, src_ptr=None, stream=None):
        _xcheck(native.lib().qs_xchg_push(self._x, C.c_void_p(int(src_ptr)) if src_ptr else None, self._s(stream)))

    def wait(self, stream=None):
        _xcheck(native.lib().qs_xchg_wait(self._x, self._s(stream)))

    def release(self, stream=None):
        _xcheck(native.lib().qs_xchg_release(self._x, self._s(stream)))

    def wait_release(self, stream=None):
        _xcheck(native.lib().qs_xchg_wait_release(self._x, self._s(stream)))

    def status(self):
        out = (C.c_int64 * 4)()
        _xcheck(native.lib().qs_xchg_status(self._x, out))
        return dict(error=int(out[0]), pushes=int(out[1]), waits=int(out[2]), releases=int(out[3]))


class ObsExchange:
    """A stepper shard + the exchange of its observation rows, overlapped with the following step.

        ex = ObsExchange(stepper, world, rank, transport="peer", wire="bf16")      # collective call when world > 1
        ex.step(actions_ptr)  ...                                                    # eager: step k, then exchange(k) on the side stream
        ex.capture(action_ptrs); ex.replay()                                         # the same, len(action_ptrs) steps as ONE HIP graph
        rows = ex.latest()                                                           # [world*T, D] of the last completed step (drains)

    The consumer side of the protocol (wait for all sources, then release the slot) runs on the side stream right behind the push:
    the stepping stream never waits for a remote rank, only for its own push two steps back (staging-buffer reuse).
    hold=True keeps the slot of the most recent step unreleased until the next step() / replay() call, so that the rows returned
    by `latest()` stay valid while the caller reads them (peers cannot overwrite a slot before it is released); hold=False
    releases a slot as soon as it has arrived (no in-place reader: the benchmark).
    """

    def __init__(self, stepper, world, rank, transport="peer", wire="bf16", group=None, peers=None, hold=True, source=None, fenced=None):
        """fenced: the fenced variant of the window transports' flag protocol (PeerExchange); source: "target" (default) - the stepper writes the rows of consecutive steps into the two staging buffers (qs_set_obs_target)
        and the exchange of step t runs under step t+1; "obs" - the rows are taken from the library's own observation buffer AFTER the
        whole qs_step, on the stepping stream: what a stepper with the device-side replay wrapper needs (default there), whose replay
        kernel restores observations into that buffer behind the step kernel (the push itself then sits between two steps; waiting
        for the peers' rows still runs on the side stream)."""
        if source is None:
            source = "obs" if getattr(stepper, "replay_on", False) else "target"
        if source not in ("target", "obs"):
            raise ValueError("source must be 'target' or 'obs'")
        if source == "obs" and transport == "fused":
            raise native.QsError("the fused exchange sends the step kernel's own rows: with the device-side replay wrapper use transport='peer' or 'rccl'")
        if source == "target" and getattr(stepper, "replay_on", False):
            raise native.QsError("a stepper with the device-side replay wrapper restores observations into qs_buffers.obs: use source='obs'")
        self.source = source
        if stepper.real_size != 4:
            raise ValueError("the exchange moves float32 observation rows (production precision)")
        if transport not in ("fused", "peer", "rccl"):
            raise ValueError("transport must be 'fused', 'peer' or 'rccl'")
        if transport == "fused" and not stepper.team:
            raise native.QsError("the fused exchange lives in the team step kernels (small batches): use transport='peer' for this batch size")
        self.st, self.world, self.rank, self.transport, self.wire, self.group = stepper, world, rank, transport, wire, group
        self.device = torch.device("cuda", stepper.device)
        self.main = torch.cuda.current_stream(self.device)
        self.comm = torch.cuda.Stream(device=self.device)
        T, D = stepper.T, stepper.obs_dim
        self.q8 = native.wire_q8_layout(stepper.cfg, D) if wire == "q8" else None
        # the endpoint also owns the two staging buffers; under "rccl" it is used for those (and its converter) only
        windows = transport in ("peer", "fused")
        if windows and peers is None and world > 1:
            # Every rank takes part in the handle exchange even if its own endpoint could not be created or exported: a rank that
            # raised before the collective would leave the others waiting in it.  All ranks then fail (or succeed) together.
            self.x, blob, err = None, None, None
            try:
                self.x = PeerExchange(T, D, world, rank, device=stepper.device, wire=wire, q8=self.q8, fenced=fenced)
                blob = self.x.export()
            except Exception as exc:   # noqa: BLE001 - re-raised below, after the collective
                err = exc
            blobs = None
            try:
                blobs = gather_window_handles(blob, world, group)
            except native.QsError as exc:
                err = err or exc
            if err is not None:
                if self.x is not None:
                    self.x.close()
                raise err
            self.x.attach(blobs)
        else:
            self.x = PeerExchange(T, D, world if windows else 1, rank if windows else 0, device=stepper.device, wire=wire, q8=self.q8, fenced=fenced)
        if windows:
            if peers is not None:          # in-process wiring (tests, one process driving several shards)
                for p in peers:
                    if p is not None and p.rank != rank:
                        self.x.attach_local(p)
        else:
            dt, wc = wire_dtype(wire), (self.x.row_bytes if wire == "q8" else D)
            self._packed = [torch.empty((T, wc), dtype=dt, device=self.device) for _ in range(2)]
            self._out = [torch.empty((world * T, wc), dtype=dt, device=self.device) for _ in range(2)]
        self.fused_on = False
        self.hold = bool(hold)
        self._pending_release = False                                 # hold: the slot of the last exchange is still ours
        self.k = 0                                                    # control steps issued so far (= pushes)
        self._stepped = [torch.cuda.Event() for _ in range(2)]
        self._done = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]
        self.graph = None
        self._graph_steps = 0
        self._capturing = False
        self._failed = None

    def _sync_main(self):
        """The stepping stream is the caller's CURRENT torch stream, resolved per call like QuadSwarmVecEnv.step without an exchange does
        (a sampler that runs under torch.cuda.stream(...) gets its steps ordered behind its own action producer and before its readers).
        A change of stream is ordered behind everything issued on the previous one."""
        if self._capturing:
            return
        cur = torch.cuda.current_stream(self.device)
        if cur != self.main:
            cur.wait_stream(self.main)
            cur.wait_stream(self.comm)
            self.main = cur

    def check(self):
        """Raise if the exchange ever timed out (sticky): a wait that gave up after QS_XCHG_TIMEOUT_MS fell through with the rows of an
        older step in the slot, an acknowledge that timed out let a peer's slot be overwritten while it may still be read.  Reads the
        device status word (a blocking copy): call it where the host synchronises anyway (episode ends, drain / close)."""
        if self._failed is None:
            st = self.status()
            if st["error"]:
                self._failed = (f"observation exchange timed out on rank {self.rank} (status {st['error']}: 1 = a peer did not release a slot, "
                                f"2 = a peer's rows did not arrive; pushes {st['pushes']}, waits {st['waits']}): rows gathered since are stale")
        if self._failed is not None:
            raise native.QsError(self._failed)

    # ---- one exchange on the side stream (eager or under capture) ----
    def _release_pending(self):
        """hold mode: hand back the slot the caller may have been reading since the previous step (readers on the stepping stream first)"""
        if self._pending_release:
            if self.transport == "fused":
                self.x.release(stream=self.main)
            else:
                self.comm.wait_stream(self.main)
                self.x.release(stream=self.comm)
            self._pending_release = False

    def _exchange(self, buf, keep=False):
        if self.source == "obs":   # rows from the library's buffer, converted / pushed on the stepping stream (the next step overwrites them)
            if self.transport == "peer":
                self.x.push(self.st.ptr("obs"), stream=self.main)
            else:
                pack_rows(self.st.tensor("obs"), self._packed[buf], stream=self.main, q8=self.q8)
            self._stepped[buf].record(self.main)
            self.comm.wait_event(self._stepped[buf])
            if self.transport == "peer":
                if keep:
                    self.x.wait(stream=self.comm)
                else:
                    self.x.wait_release(stream=self.comm)
            else:
                with torch.cuda.stream(self.comm):
                    if self.world > 1 or dist.is_initialized():
                        dist.all_gather_into_tensor(self._out[buf], self._packed[buf], group=self.group)
                    else:
                        self._out[buf].copy_(self._packed[buf])
            return
        if self.transport == "peer":
            self.x.push(self.x.staging_ptr(buf), stream=self.comm)
            if keep:
                self.x.wait(stream=self.comm)           # the slot stays ours until _release_pending()
            else:
                self.x.wait_release(stream=self.comm)   # one launch
        else:
            with torch.cuda.stream(self.comm):
                pack_rows(self.x.staging(buf), self._packed[buf], stream=self.comm, q8=self.q8)
                if self.world > 1 or dist.is_initialized():
                    dist.all_gather_into_tensor(self._out[buf], self._packed[buf], group=self.group)
                else:
                    self._out[buf].copy_(self._packed[buf])

    def _fuse(self):
        """switch the stepper to the fused push (after the windows are wired and, in bench.py, self-checked through the push kernel)"""
        if not self.fused_on:
            self.drain()
            torch.cuda.synchronize(self.device)
            self.st.set_obs_target(None)
            self.st.set_obs_exchange(self.x._x, auto_ack=not self.hold)
            self.fused_on = True

    def pause(self):
        """steps issued directly on the stepper from now on do not exchange their rows (bench.py: the independent-shards figure)"""
        self.drain()
        self.st.set_obs_target(None)
        if self.fused_on:
            torch.cuda.synchronize(self.device)
            self.st.set_obs_exchange(None)
            self.fused_on = False

    def _one(self, actions_ptr, buf, keep=False):
        if self.transport == "fused":   # the step launch is the exchange
            self._fuse()
            self.st.step(actions_ptr, stream=self.main)
            if self.hold:
                self.x.wait(stream=self.main)        # (reader mode: the slot stays ours until _release_pending())
            return
        if self._used[buf]:
            self.main.wait_event(self._done[buf])                    # the push that read staging[buf] two steps ago has finished
        if self.source == "obs":
            self.st.step(actions_ptr, stream=self.main)              # (step kernel + replay kernel)
            self._exchange(buf, keep)
            self._done[buf].record(self.comm)
            self._used[buf] = True
            return
        self.st.set_obs_target(self.x.staging_ptr(buf))
        self.st.step(actions_ptr, stream=self.main)
        self._stepped[buf].record(self.main)
        self.comm.wait_event(self._stepped[buf])
        self._exchange(buf, keep)
        self._done[buf].record(self.comm)
        self._used[buf] = True

    def step(self, actions_ptr):
        """control step k (eager): observations -> staging[k & 1]; exchange(k) on the side stream under step k+1"""
        if self._failed is not None:
            raise native.QsError(self._failed)
        self._sync_main()
        self._release_pending()
        self._one(actions_ptr, self.k & 1, keep=self.hold)
        self._pending_release = self.hold and self.transport in ("peer", "fused")
        self.k += 1

    def align(self, actions_ptr):
        """make the number of issued steps even (what replay() needs) with at most one eager step"""
        if self.k & 1:
            self.step(actions_ptr)

    def reset(self):
        """reset all envs; the first observation rows are exchanged like a step's"""
        self._sync_main()
        self._release_pending()
        self.drain()
        if self.transport == "fused":   # the reset kernel has no epilogue: its rows go out through the push kernel, from the library's own buffer
            self._fuse()
            self.st.reset(stream=self.main)
            self.x.push(self.st.ptr("obs"), stream=self.main)
            if self.hold:
                self.x.wait(stream=self.main)
            else:
                self.x.wait_release(stream=self.main)
            self._pending_release = self.hold
            self.k += 1
            return
        buf = self.k & 1
        if self.source == "obs":
            self.st.reset(stream=self.main)
        else:
            self.st.set_obs_target(self.x.staging_ptr(buf))
            self.st.reset(stream=self.main)
            self._stepped[buf].record(self.main)
            self.comm.wait_event(self._stepped[buf])
        self._exchange(buf, keep=self.hold)
        self._pending_release = self.hold and self.transport == "peer"
        self._done[buf].record(self.comm)
        self._used[buf] = True
        self.k += 1

    # ---- a segment as one HIP graph ----
    def capture(self, action_ptrs):
        """Record len(action_ptrs) control steps (an even number: the staging buffers alternate across replays) of
        [step -> exchange on the side stream] into one HIP graph.  Runs eager steps first until at least two have been issued and
        their number is even (kernel modules get loaded outside the capture; graphs start on staging[0])."""
        n = len(action_ptrs)
        if n < 2 or n % 2:
            raise ValueError("capture an even number of steps")
        if self.transport == "rccl" and dist.is_available() and dist.is_initialized():
            # torch's ProcessGroupNCCL keeps every collective it issues on its watchdog thread's list, also one recorded under stream
            # capture; the watchdog's later query of that captured event raises hipErrorCapturedEvent and aborts the process
            # (seen once in five runs of bench.py --force-gather --transport rccl: profiles/r03x_rccl_capture_abort.txt)
            raise native.QsError("the rccl transport is not captured into a HIP graph: step it eagerly (ObsExchange.step)")
        self.drain()
        while self.k < 2 or self.k & 1:
            self.step(action_ptrs[0])
            self.drain()
        self._release_pending()
        self.drain()
        if self.transport == "fused":
            self._fuse()   # outside the capture (it synchronises the device): also when the eager pre-steps above did not run - k already
                           # even after pause() or self_check() - and the stepper is still on its plain obs target
        self.graph = torch.cuda.CUDAGraph()
        self._used = [False, False]
        with torch.cuda.graph(self.graph, stream=self._capture_stream()):
            cap = torch.cuda.current_stream(self.device)
            saved, self.main = self.main, cap
            self._capturing = True
            try:
                for t in range(n):
                    assert self.transport != "fused" or self.fused_on
                    if self.transport == "fused" and self.hold:   # reader mode inside a segment: every slot but the last is released at once
                        self.st.step(action_ptrs[t], stream=self.main)
                        self.x.wait(stream=self.main)
                        if t < n - 1:
                            self.x.release(stream=self.main)
                        continue
                    self._one(action_ptrs[t], t & 1, keep=self.hold and t == n - 1)   # hold: the last slot of a segment stays ours
                cap.wait_stream(self.comm)                             # join: the graph ends when its last exchange has
            finally:
                self.main = saved
                self._capturing = False
        self._used = [False, False]                                    # everything recorded is ordered by the graph launch itself
        self._graph_steps = n
        if self.transport != "fused" and self.source == "target":
            self.st.set_obs_target(self.x.staging_ptr(self.k & 1))
        return self

    def _capture_stream(self):
        if not hasattr(self, "_cap"):
            self._cap = torch.cuda.Stream(device=self.device)
        return self._cap

    def replay(self):
        """replay the captured segment on the stepping stream (stream-ordered behind everything issued so far on both streams)"""
        if self.k & 1:
            raise native.QsError("replay() needs an even number of steps issued before it (staging parity)")
        if self._failed is not None:
            raise native.QsError(self._failed)
        self._sync_main()
        self._release_pending()
        self.main.wait_stream(self.comm)
        for b in (0, 1):
            self._used[b] = False
        with torch.cuda.stream(self.main):
            self.graph.replay()
        self._pending_release = self.hold and self.transport in ("peer", "fused")
        self.k += self._graph_steps

    # ---- start-up self-check of the peer-store transport ----
    def self_check(self, rounds=6):
        """Exchange `rounds` (even) synthetic row sets whose content every rank can compute for every rank, and compare what arrived
        with it: exercises both window slots, the flow control and - on a multi-GPU node - the visibility of remote stores to local
        readers, with no collective involved.  Returns (ok, reason).  Call it on every rank at the same point."""
        if self.transport == "rccl":
            return True, "no windows to check"
        if rounds % 2:
            rounds += 1
        self._release_pending()
        self.drain()
        T, D, W = self.st.T, self.st.obs_dim, self.world
        base = torch.arange(T * D, device=self.device, dtype=torch.float32).reshape(T, D)
        iv = {"f32": torch.int32, "bf16": torch.int16, "q8": torch.uint8}[self.wire]
        ok, why = True, ""
        for i in range(rounds):
            buf = i & 1
            with torch.cuda.stream(self.comm):
                self.x.staging(buf).copy_(torch.sin(base * (0.37 + 0.01 * self.rank) + float(i)) * (1.0 + self.rank))
            self.x.push(self.x.staging_ptr(buf), stream=self.comm)
            self.x.wait(stream=self.comm)
            with torch.cuda.stream(self.comm):
                got = self.x.gathered((self.k + i + 1) & 1).clone()
            self.x.release(stream=self.comm)
            self.comm.synchronize()
            want = torch.cat([quantize_rows_reference(torch.sin(base * (0.37 + 0.01 * r) + float(i)) * (1.0 + r), self.wire, self.q8) for r in range(W)])
            if not torch.equal(got.view(iv), want.view(iv)):
                bad = (got.view(iv) != want.view(iv)).reshape(W, -1).any(dim=1).nonzero().flatten().tolist()
                ok, why = False, f"round {i}: rows of rank(s) {bad} differ from what they sent"
                break
        st = self.x.status()
        if st["error"]:
            ok, why = False, (why + "; " if why else "") + f"exchange status {st['error']} (1 = ack timeout, 2 = arrive timeout)"
        self.k += rounds   # pushes so far (even: the staging parity of the steps is unchanged)
        return ok, why

    def verify(self):
        """Compare the gathered rows of the most recent step with an independent gather of the same rows - every rank's float32 rows,
        converted to the wire type by the library's converter, through torch.distributed (RCCL) - and return (ok, reason).  Unlike
        self_check() this goes through whatever produced the rows in the windows: the step kernel's fused epilogue on the real
        topology included.  Collective: call it on every rank at the same step (needs hold=True, or a quiet exchange: nothing may
        overwrite the slot meanwhile).  bench.py runs it before and after its timed region, sf_env when a window transport is opted into."""
        # the part that can fail on ONE rank (a sticky timeout, a bad window) runs first and is agreed on by all ranks BEFORE anyone enters
        # the all-gather: a rank that raised here while the others sat in the collective would leave mismatched collectives behind
        got = mine = None
        local_error = ""
        try:
            got = self.latest().clone()
            mine = torch.empty((self.st.T, got.shape[1]), dtype=got.dtype, device=self.device)
            pack_rows(self.local_rows(), mine, stream=self.main, q8=self.q8)
        except Exception as exc:   # noqa: BLE001 - reported as the reason, on every rank
            local_error = f"{type(exc).__name__}: {exc}"
        if self.world > 1:
            flag = torch.tensor([0 if local_error else 1], device=self.device, dtype=torch.int32)
            with torch.cuda.stream(self.main):
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            self.main.synchronize()
            if int(flag.item()) == 0:
                return False, local_error or "another rank could not read its gathered rows"
        elif local_error:
            return False, local_error
        dt = got.dtype
        if self.world > 1:
            want = torch.empty_like(got)
            with torch.cuda.stream(self.main):
                dist.all_gather_into_tensor(want, mine, group=self.group)
        else:
            want = mine
        self.main.synchronize()
        iv = {torch.float32: torch.int32, torch.bfloat16: torch.int16, torch.uint8: torch.uint8}[dt]
        if torch.equal(got.view(iv), want.view(iv)):
            return True, ""
        bad = (got.view(iv) != want.view(iv)).reshape(self.world, -1).any(dim=1).nonzero().flatten().tolist()
        return False, f"gathered rows of rank(s) {bad} differ from what an RCCL all-gather of the same rows delivers (step {self.k})"

    # ---- results ----
    def drain(self):
        """the stepping stream waits for every exchange issued so far"""
        self.main.wait_stream(self.comm)

    def latest(self, check=False):
        """gathered rows [world*T, D] (wire dtype) of the most recent step, valid on the stepping stream.  check=True also reads the
        exchange's status word (a blocking copy) and raises if a wait ever timed out - see check()."""
        if self._failed is not None:
            raise native.QsError(self._failed)
        self.drain()
        if check:
            self.check()
        slot = self.k & 1 if self.transport != "rccl" else (self.k - 1) & 1   # windows: slot = push sequence number & 1 (1-based)
        return self.x.gathered(slot) if self.transport != "rccl" else self._out[slot]

    def latest_f32(self, out=None):
        """the gathered rows of the most recent step as float32 [world*T, D] (bf16 widened, q8 dequantised: qs_obs_unpack_rows)"""
        g = self.latest()
        with torch.cuda.stream(self.main):
            return unpack_rows(g, self.st.obs_dim, self.wire, self.q8, out=out, stream=self.main)

    def local_rows(self):
        """this rank's float32 rows of the most recent step (the staging buffer the stepper wrote; fused: the library's own `obs`)"""
        if self.transport == "fused" or self.source == "obs":
            return self.st.tensor("obs")
        return self.x.staging((self.k - 1) & 1)

    def status(self):
        return self.x.status() if self.transport != "rccl" else dict(error=0, pushes=self.k, waits=self.k, releases=self.k)

    def close(self):
        torch.cuda.synchronize(self.device)
        self.st.set_obs_target(None)
        if self.fused_on:
            self.st.set_obs_exchange(None)
            self.fused_on = False
        self.graph = None
        self.x.close()


class ObsGather:
    """Round-1/2 transport, kept for comparison (bench.py --transport torch): per-step torch.distributed all-gather of the float32
    observation tensor, optionally double-buffered on RCCL's own stream.  Costs ~30 us of host time per step."""

    def __init__(self, local_obs, group=None, overlap=False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = local_obs
        self.overlap = overlap
        nbuf = 2 if overlap else 1
        shape = (self.world * local_obs.shape[0],) + tuple(local_obs.shape[1:])
        self.out = [torch.empty(shape, dtype=local_obs.dtype, device=local_obs.device) for _ in range(nbuf)]
        self.stage = [torch.empty_like(local_obs) for _ in range(nbuf)] if overlap else None
        self.work = [None] * nbuf
        self.t = 0

    def _all_gather(self, out, src, async_op):
        try:
            return dist.all_gather_into_tensor(out, src, group=self.group, async_op=async_op)
        except (RuntimeError, NotImplementedError):   # backend without the fused op: list form, same result
            chunks = list(out.chunk(self.world, dim=0))
            return dist.all_gather(chunks, src, group=self.group, async_op=async_op)

    def gather(self):
        """Call right after the step that produced `local_obs`.  Returns the gathered tensor (overlap=False) or the
        index of the in-flight buffer (overlap=True; fetch with result(idx))."""
        if not self.overlap:
            self._all_gather(self.out[0], self.local, async_op=False)
            return self.out[0]
        k = self.t & 1
        if self.work[k] is not None:
            self.work[k].wait()                 # buffer k was used at step t-2
        self.stage[k].copy_(self.local)         # snapshot: the stepper overwrites its obs buffer at the next step
        self.work[k] = self._all_gather(self.out[k], self.stage[k], async_op=True)
        self.t += 1
        return k

    def result(self, k):
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        return self.out[k]

    def drain(self):
        for k in range(len(self.work)):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None

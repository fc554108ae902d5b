"""One process per GPU: slab partition + face-neighbour halo exchange for the explicit path.

Replaces what the MPI variant gets from deal.II (src_mpi/):
  current_solution.update_ghost_values()  src_mpi/claw.cc:793, src_mpi/limiter.cc:232
  Utilities::MPI::min(global_dt)          src_mpi/claw.cc:579
  right_hand_side.l2_norm()               src_mpi/claw.cc:777
  right_hand_side.compress(add)           src_mpi/assemble_explicit.cc:580  -- not needed: faces on a
      partition boundary are integrated by both owners with the same integrating side (bit-identical flux).
The transport is torch.distributed point-to-point (backend "nccl" = RCCL over xGMI on the GPU box;
"gloo" with host staging in CPU tests).  The data path has exactly one exchange per RK stage (two
when the TVB limiter needs the neighbours' fresh cell averages) and one 8-byte all-reduce per step.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import lib
from .claw import ConservationLaw


class _DevPtr:
    """Wraps a raw device address as a torch tensor (no copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _device_view(ptr, n, device):
    return torch.as_tensor(_DevPtr(ptr, n), device=device)


class HaloExchange:
    """Point-to-point exchange of per-cell records between neighbouring ranks.

    send_offsets / recv_offsets (size world+1) come from Mesh.partition: cells to send to rank r are
    send_cells[send_offsets[r]:send_offsets[r+1]]; ghost cells received from r occupy
    [recv_offsets[r], recv_offsets[r+1]) of the ghost range."""

    def __init__(self, send_offsets, recv_offsets, device):
        self.so = [int(v) for v in send_offsets]
        self.ro = [int(v) for v in recv_offsets]
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.device = device
        self.host_staging = dist.get_backend() != "nccl" and device.type == "cuda"
        # RCCL: one grouped collective per exchange (all_to_all_single = grouped ncclSend/ncclRecv with the
        # per-peer counts below) instead of 2 x peers python-level P2P ops; DFLO_HALO=p2p forces the latter
        import os
        self.use_a2a = os.environ.get("DFLO_HALO", "a2a") != "p2p"
        self.send_counts = [self.so[r + 1] - self.so[r] for r in range(self.world)]
        self.recv_counts = [self.ro[r + 1] - self.ro[r] for r in range(self.world)]
        self.peers = [r for r in range(self.world) if r != self.rank and (self.so[r + 1] > self.so[r] or self.ro[r + 1] > self.ro[r])]

    def exchange_async(self, send, recv, width):
        """Start the exchange and return a handle; kernels launched before wait() run concurrently with the
        transfer (the collective runs on the process group's own stream)."""
        if not self.peers:
            return _Pending()
        if not self.use_a2a:          # grouped point-to-point has no cheap asynchronous form here: do it now
            self.exchange(send, recv, width)
            return _Pending()
        n_s, n_r = self.so[-1] * width, self.ro[-1] * width
        rc, sc = [c * width for c in self.recv_counts], [c * width for c in self.send_counts]
        if self.host_staging:
            s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            work = dist.all_to_all_single(r[:n_r], s[:n_s], rc, sc, async_op=True)
            return _Pending(work, lambda: recv.copy_(r))
        return _Pending(dist.all_to_all_single(recv[:n_r], send[:n_s], rc, sc, async_op=True))

    def exchange(self, send, recv, width):
        """send: [n_send*width] tensor, recv: [n_ghost*width] tensor (both on self.device)."""
        if not self.peers:
            return
        if self.host_staging:  # gloo with device buffers (tests): stage through the host
            s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        else:
            s, r = send, recv
        if self.use_a2a:
            n_s, n_r = self.so[-1] * width, self.ro[-1] * width
            try:
                dist.all_to_all_single(r[:n_r], s[:n_s], [c * width for c in self.recv_counts],
                                       [c * width for c in self.send_counts])
            except RuntimeError:
                # a backend that refuses uneven all-to-all: every rank sees the same error at the same call,
                # so all of them switch to grouped point-to-point together
                self.use_a2a = False
        if not self.use_a2a:
            ops = []
            for p in self.peers:
                if self.ro[p + 1] > self.ro[p]:
                    ops.append(dist.P2POp(dist.irecv, r[self.ro[p] * width:self.ro[p + 1] * width], p))
            for p in self.peers:
                if self.so[p + 1] > self.so[p]:
                    ops.append(dist.P2POp(dist.isend, s[self.so[p] * width:self.so[p + 1] * width], p))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.host_staging:
            recv.copy_(r)


class _Pending:
    """Handle of an exchange in flight: wait() orders the caller's stream after the transfer (RCCL: a stream
    dependency, no host block) and, with host staging, copies the received data to the device buffer."""

    def __init__(self, work=None, after=None):
        self.work, self.after = work, after

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.after is not None:
            self.after()


def _on_main_stream(method):
    """Run a method of DistributedConservationLaw with its main stream as torch's current stream."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with torch.cuda.stream(self.main_stream):
            return method(self, *args, **kwargs)
    return wrapped


class DistributedConservationLaw:
    """ConservationLaw on the slab of this rank (torch.distributed must be initialised)."""

    def __init__(self, global_mesh, parameters, device_index=0):
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.global_mesh = global_mesh
        self.mesh = global_mesh.partition(self.world, self.rank)
        self.parameters = parameters
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.claw = ConservationLaw(self.mesh, parameters, device=device_index)
        # One explicit stream for the engine's kernels and (as torch's current stream inside every method below) for the
        # collectives, so that the process group orders its transfers against exactly that stream.  (The default
        # stream would also work, through the implicit synchronisation of the legacy null stream -- and would
        # serialise against whatever else the process runs there.)
        self.main_stream = torch.cuda.Stream(device=self.device)
        self.claw.set_stream(self.main_stream.cuda_stream)
        # second stream for the halo traffic: the rim shards are advanced first, their cells travel while the
        # interior shards are computed (DFLO_OVERLAP=0 switches back to the serial order)
        import os
        # DFLO_OVERLAP: 1 (default) the exchange is an asynchronous collective on the process group's stream, everything
        # else stays on one stream; 2: pack / exchange / unpack on a second stream of our own; 0: serial order
        self.overlap = {"0": 0, "2": 2}.get(os.environ.get("DFLO_OVERLAP", "1"), 1)
        self.comm_stream = torch.cuda.Stream(device=self.device)

        send_cells, so, ro = self.mesh.comm
        self.halo = HaloExchange(so, ro, self.device)
        sc = np.ascontiguousarray(send_cells, dtype=np.int32)
        self.claw._chk(lib.dflo_hip_set_send_cells(self.claw._h, len(sc), _lib.iptr(sc)))
        ndof = self.claw.dofs_per_cell
        self.n_send = len(sc)
        self.n_ghost = self.mesh.n_cells - self.mesh.n_owned
        self.send_u = torch.empty(max(self.n_send, 1) * ndof, dtype=torch.float64, device=self.device)
        self.recv_u = torch.empty(max(self.n_ghost, 1) * ndof, dtype=torch.float64, device=self.device)
        self.send_a = torch.empty(max(self.n_send, 1) * 4, dtype=torch.float64, device=self.device)
        self.recv_a = torch.empty(max(self.n_ghost, 1) * 4, dtype=torch.float64, device=self.device)
        self.ndof = ndof
        self.n_rk = self.claw.n_rk
        self.tvb = parameters.limiter == "TVB"
        # KXRCF indicator: reads the neighbours' unlimited DoFs of the new stage, so the ghost cells are refreshed once
        # more between update and limiter (what update_ghost_values before compute_shock_indicator does in the MPI
        # variant); that path runs without the rim/interior overlap
        self.kxrcf = self.tvb and parameters.shock_indicator != "limiter"
        if self.kxrcf:
            self.overlap = False
        # device-resident {dt, elapsed time, raw CFL minimum}: lets the step loop run without host round trips
        dtp, resp = C.c_void_p(), C.c_void_p()
        self.claw._chk(lib.dflo_hip_scalar_ptrs(self.claw._h, C.byref(dtp), C.byref(resp)))
        self.dt_dev = _device_view(dtp.value, 4, self.device)
        self.elapsed_time = 0.0
        self.n_dofs_owned = self.mesh.n_owned * ndof
        self.n_dofs_global = global_mesh.n_cells * ndof

    # ---- data movement
    def owned_slice_of_global(self, u_global):
        gid = np.asarray(self.mesh.global_ids)
        return np.ascontiguousarray(np.asarray(u_global).reshape(self.global_mesh.n_cells, self.ndof)[gid]).reshape(-1)

    @_on_main_stream
    def set_initial_condition(self, u_global):
        self._join()
        self.claw.set_initial_condition(self.owned_slice_of_global(u_global))

    @_on_main_stream
    def exchange_solution(self):
        c = self.claw
        c._chk(lib.dflo_hip_pack_send(c._h, C.c_void_p(self.send_u.data_ptr())))
        self.halo.exchange(self.send_u, self.recv_u, self.ndof)
        c._chk(lib.dflo_hip_unpack_ghost(c._h, C.c_void_p(self.recv_u.data_ptr())))

    @_on_main_stream
    def exchange_averages(self):
        c = self.claw
        c._chk(lib.dflo_hip_pack_send_avg(c._h, C.c_void_p(self.send_a.data_ptr())))
        self.halo.exchange(self.send_a, self.recv_a, 4)
        c._chk(lib.dflo_hip_unpack_ghost_avg(c._h, C.c_void_p(self.recv_a.data_ptr())))

    # ---- time stepping
    @_on_main_stream
    def compute_time_step(self):
        self.claw.elapsed_time = self.elapsed_time
        dt = self.claw.compute_time_step()
        t = torch.tensor([dt], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)  # Utilities::MPI::min, src_mpi/claw.cc:579
        return float(t.item())

    def _stage(self, rk, dt):
        """One RK stage with its halo exchange(s)."""
        c = self.claw
        if not self.overlap:
            c._chk(lib.dflo_hip_stage_update(c._h, rk, dt))
            if self.kxrcf:
                self.exchange_solution()      # ghost DoFs and their averages of the unlimited stage
            elif self.tvb:
                self.exchange_averages()
            c._chk(lib.dflo_hip_stage_limit(c._h))
            self.exchange_solution()
            return
        if self.overlap == 1:
            # rim shards first; their cells travel while the interior shards are computed
            c._chk(lib.dflo_hip_stage_open(c._h, rk, dt))
            c._chk(lib.dflo_hip_stage_update_part(c._h, 1))
            if self.tvb:   # the limiter of the rim cells needs the neighbours' fresh means
                c._chk(lib.dflo_hip_pack_send_avg(c._h, C.c_void_p(self.send_a.data_ptr())))
                pa = self.halo.exchange_async(self.send_a, self.recv_a, 4)
                c._chk(lib.dflo_hip_stage_update_part(c._h, 2))          # overlaps the (small) exchange of the means
                pa.wait()
                c._chk(lib.dflo_hip_unpack_ghost_avg(c._h, C.c_void_p(self.recv_a.data_ptr())))
                c._chk(lib.dflo_hip_stage_limit_part(c._h, 1))
                c._chk(lib.dflo_hip_pack_send(c._h, C.c_void_p(self.send_u.data_ptr())))
                pu = self.halo.exchange_async(self.send_u, self.recv_u, self.ndof)
                c._chk(lib.dflo_hip_stage_limit_part(c._h, 2))
            else:
                c._chk(lib.dflo_hip_stage_limit_part(c._h, 1))
                c._chk(lib.dflo_hip_pack_send(c._h, C.c_void_p(self.send_u.data_ptr())))
                pu = self.halo.exchange_async(self.send_u, self.recv_u, self.ndof)
                c._chk(lib.dflo_hip_stage_update_part(c._h, 2))
                c._chk(lib.dflo_hip_stage_limit_part(c._h, 2))
            c._chk(lib.dflo_hip_stage_finish(c._h))
            pu.wait()
            c._chk(lib.dflo_hip_unpack_ghost(c._h, C.c_void_p(self.recv_u.data_ptr())))
            return
        main, comm = self.main_stream, self.comm_stream
        mainp, commp = C.c_void_p(main.cuda_stream), C.c_void_p(comm.cuda_stream)
        c._chk(lib.dflo_hip_stage_rim(c._h, rk, dt, mainp))                      # rim shards on the main stream
        with torch.cuda.stream(comm):
            if self.tvb:   # the limiter of the rim cells needs the neighbours' fresh means first
                c._chk(lib.dflo_hip_stage_rim_send(c._h, commp, 1, None, C.c_void_p(self.send_a.data_ptr())))
                self.halo.exchange(self.send_a, self.recv_a, 4)
                c._chk(lib.dflo_hip_stage_rim_send(c._h, commp, 2, C.c_void_p(self.recv_a.data_ptr()),
                                                   C.c_void_p(self.send_u.data_ptr())))
            else:
                c._chk(lib.dflo_hip_stage_rim_send(c._h, commp, 0, None, C.c_void_p(self.send_u.data_ptr())))
            self.halo.exchange(self.send_u, self.recv_u, self.ndof)
            c._chk(lib.dflo_hip_stage_rim_recv(c._h, commp, C.c_void_p(self.recv_u.data_ptr())))
        c._chk(lib.dflo_hip_stage_interior(c._h))                                # concurrently with the exchange

    def _join(self):
        if self.overlap == 2:
            self.claw._chk(lib.dflo_hip_stage_join(self.claw._h))

    @_on_main_stream
    def iterate_explicit(self, dt):
        c = self.claw
        for rk in range(self.n_rk):
            self._stage(rk, dt)
        self._join()
        c.end_step()
        self.elapsed_time += dt

    @_on_main_stream
    def advance(self, n_steps):
        """n_steps x {compute_time_step; iterate_explicit} with dt resident on the device: per step one
        8-byte all-reduce(min) (Utilities::MPI::min, src_mpi/claw.cc:579) and no host synchronisation."""
        c = self.claw
        dt0 = self.compute_time_step()                 # host value for the first step only
        nccl = dist.get_backend() == "nccl"
        for step in range(n_steps):
            for rk in range(self.n_rk):
                self._stage(rk, dt0 if step == 0 else -1.0)
            c.end_step()
            # the last stage left this rank's raw CFL minimum in dt_dev[2]
            if nccl:
                dist.all_reduce(self.dt_dev[2:3], op=dist.ReduceOp.MIN)
            else:
                t = self.dt_dev[2:3].cpu()
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                self.dt_dev[2:3].copy_(t)
            c._chk(lib.dflo_hip_apply_dt_rules(c._h))
        self._join()
        self.elapsed_time = float(self.dt_dev[1].item())
        return self.elapsed_time

    @_on_main_stream
    def residual_norms(self):
        """(||rhs|| of the first stage, of the last stage) of the step just done, summed over the ranks --
        right_hand_side.l2_norm() of src_mpi/claw.cc:777 (a printed diagnostic; reduced here on demand, not per stage)."""
        self._join()
        dtp, resp = C.c_void_p(), C.c_void_p()
        self.claw._chk(lib.dflo_hip_scalar_ptrs(self.claw._h, C.byref(dtp), C.byref(resp)))
        r = _device_view(resp.value, 4, self.device).clone()
        if dist.get_backend() != "nccl":
            r = r.cpu()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        r = r.cpu().numpy()
        return float(np.sqrt(r[0])), float(np.sqrt(r[self.n_rk - 1]))

    @_on_main_stream
    def gather_solution(self):
        """Owned DoFs of all ranks assembled in the global cell order (on every rank; test helper)."""
        self._join()
        u = self.claw.current_solution.reshape(self.mesh.n_cells, self.ndof)[: self.mesh.n_owned]
        gid = np.asarray(self.mesh.global_ids)[: self.mesh.n_owned]
        parts = [None] * self.world
        dist.all_gather_object(parts, (gid, u))
        out = np.empty((self.global_mesh.n_cells, self.ndof))
        for g, v in parts:
            out[g] = v
        return out.reshape(-1)

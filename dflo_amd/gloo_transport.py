"""A host-staged transport for the native multi-device driver's bring-your-own-transport entry
(dflo_hip_multi_create_rank_custom): the callbacks copy the device buffers through the host and move them with
torch.distributed (gloo).  It exists so that the one-process-per-GPU schedule can be run by several processes on ONE
GPU, which RCCL refuses -- the tests and `DFLO_BENCH_TRANSPORT=gloo python -m torch.distributed.run ... bench.py` use
it; production runs use RCCL (MultiConservationLaw.for_rank).  Slow by construction (every exchange drains the stream)."""
import sys

import torch
import torch.distributed as dist


class _DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _view(ptr, nbytes, device):
    return torch.as_tensor(_DevPtr(ptr, nbytes // 8), device=device)


def make_callbacks(device="cuda:0"):
    """(exchange, allreduce) for MultiConservationLaw.for_rank_custom; torch.distributed must be initialised (gloo)."""
    dev = torch.device(device)

    def _stream(stream):
        # the driver's comm stream (hipStream_t handed over as an integer): the staging copies are enqueued on it, behind the
        # pack kernel and ahead of whatever the driver enqueues next -- no reliance on the legacy null stream or on torch's
        # current device
        return torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev)

    def exchange(user, n_peers, peer, send_ptr, send_bytes, recv_ptr, recv_bytes, stream):
        try:
            st = _stream(stream)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                st.synchronize()            # the pack kernel has filled the send buffers
                ops, back = [], []
                for i in range(n_peers):
                    if recv_bytes[i]:
                        host = torch.empty(recv_bytes[i] // 8, dtype=torch.float64)
                        back.append((host, recv_ptr[i], recv_bytes[i]))
                        ops.append(dist.P2POp(dist.irecv, host, peer[i]))
                    if send_bytes[i]:
                        ops.append(dist.P2POp(dist.isend, _view(send_ptr[i], send_bytes[i], dev).cpu(), peer[i]))
                for w in dist.batch_isend_irecv(ops) if ops else []:
                    w.wait()
                for host, ptr, nb in back:
                    _view(ptr, nb, dev).copy_(host)
                st.synchronize()            # pageable host memory: the copies must not outlive `host`
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            print("exchange callback:", e, file=sys.stderr)
            return 1

    def allreduce(user, values, n, op, stream):
        try:
            st = _stream(stream)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                st.synchronize()
                v = _view(values, 8 * n, dev)
                h = v.cpu()
                dist.all_reduce(h, op=[dist.ReduceOp.MIN, dist.ReduceOp.SUM, dist.ReduceOp.MAX][op])
                v.copy_(h)
                st.synchronize()
            return 0
        except Exception as e:
            print("allreduce callback:", e, file=sys.stderr)
            return 1

    return exchange, allreduce

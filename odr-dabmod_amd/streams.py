"""Multi-GPU harness: one process per GPU, each modulating its own independent
stream of transmission frames (frames are independent units, so the
path shards with NO data-path collective).  torch.distributed (backend "nccl" =
RCCL on ROCm, "gloo" in CPU tests) is used only to bracket the timed region and
to combine the ranks' clocks."""
import os
import sys
import time


class StreamGroup:
    def __init__(self, backend=None, device_index=None):
        import torch
        import torch.distributed as dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # the GPU this rank drives (default: its local rank)
        self.device_index = self.local_rank if device_index is None else int(device_index)
        self._dist = dist
        self._torch = torch
        self.backend = backend
        # DABGPU_FORCE_DIST=1 initialises the process group for a single rank too (exercises the RCCL path
        # on a one-GPU box)
        self._collective = self.world > 1 or os.environ.get("DABGPU_FORCE_DIST") == "1"
        if self._collective and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.device_index)
            # RCCL prints its version banner on the C-level stdout whenever it first creates a communicator -- at
            # init, at the first collective of a kind, or as late as teardown.  The bench's stdout carries exactly
            # one JSON line, so for the life of the group file descriptor 1 points at stderr and the line goes out
            # through a private duplicate of the real stdout (emit()).
            sys.stdout.flush()
            self._out_fd = os.dup(1)
            os.dup2(2, 1)
            dist.init_process_group(backend, **kw)
            dist.barrier()

    def emit(self, text):
        """Write one line to the process's real stdout (see __init__)."""
        fd = getattr(self, "_out_fd", None)
        if fd is None:
            print(text, flush=True)
        else:
            os.write(fd, (text + "\n").encode())

    def barrier(self):
        if self._collective:
            self._dist.barrier()

    def max_over_ranks(self, seconds):
        """The job's elapsed time is the slowest rank's."""
        if not self._collective:
            return float(seconds)
        dev = "cuda:%d" % self.device_index if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([float(seconds)], dtype=self._torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, value):
        """The smallest of the ranks' integers (the batch every rank can hold: weak scaling keeps per-GPU work equal)."""
        if not self._collective:
            return int(value)
        dev = "cuda:%d" % self.device_index if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([int(value)], dtype=self._torch.int64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN)
        return int(t.item())

    def stream_seed(self, base=42):
        """Every rank modulates a different stream (different synthetic input)."""
        return base + self.rank

    def timed(self, fn, steps, sync):
        """barrier + device sync, `steps` calls of fn, device sync + barrier; MAX over ranks."""
        self.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def job_frames_per_second(self, frames_per_step_per_gpu, steps, seconds):
        """Whole-job throughput: every rank processed frames_per_step_per_gpu * steps frames."""
        return self.world * frames_per_step_per_gpu * steps / seconds

    def gather_to_root(self, t, root=0):
        """OPTIONAL final gather of the ranks' IQ (north_star: "RCCL over xGMI only for an optional
        final IQ gather"): every rank's tensor `t` lands on `root`, in rank order.  Never on the modulation path --
        frames are independent and each stream's sink may just as well sit behind its own GPU.  Over xGMI every
        non-root rank owns one point-to-point link to the root (~153 GB/s each way), so the root receives
        (N - 1) * t.nbytes over N - 1 links in parallel: the gather is bound per link by t.nbytes / 153 GB/s, and
        on the root by its HBM write rate only.  Returns the list of N tensors on the root, None elsewhere."""
        torch, dist = self._torch, self._dist
        if not self._collective:
            return [t]
        cplx = t.is_complex()
        if cplx:
            t = torch.view_as_real(t)            # (RCCL has no complex element type: the same bytes as float pairs)
        dev = t.device
        if self.backend == "gloo" and dev.type != "cpu":
            t = t.cpu()                          # (gloo gathers host tensors only: ranks that share a GPU in the tests)
        out = [torch.empty_like(t) for _ in range(self.world)] if self.rank == root else None
        dist.gather(t.contiguous(), out, dst=root)
        if out is not None and out[0].device != dev:
            out = [o.to(dev) for o in out]
        if out is not None and cplx:
            out = [torch.view_as_complex(o) for o in out]
        return out

    def close(self):
        if self._collective and self._dist.is_initialized():
            self._dist.destroy_process_group()
        # (fd 1 stays on stderr until the process exits: RCCL may still print while it unloads)

"""Multi-GPU harness: one process per GPU, each modulating its own independent
stream of transmission frames (frames are independent units, so the
path shards with NO data-path collective).  torch.distributed (backend "nccl" =
RCCL on ROCm, "gloo" in CPU tests) is used only to bracket the timed region and
to combine the ranks' clocks."""
import contextlib
import os
import sys
import time


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on the C-level stdout when a communicator is created; the bench's
    stdout carries exactly one JSON line, so the banner is sent to stderr."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class StreamGroup:
    def __init__(self, backend=None):
        import torch
        import torch.distributed as dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
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
                kw["device_id"] = torch.device("cuda", self.local_rank)
            with _stdout_to_stderr():
                dist.init_process_group(backend, **kw)
                dist.barrier()                    # communicator creation (and its banner) happens here at the latest

    def barrier(self):
        if self._collective:
            self._dist.barrier()

    def max_over_ranks(self, seconds):
        """The job's elapsed time is the slowest rank's."""
        if not self._collective:
            return float(seconds)
        dev = "cuda:%d" % self.local_rank if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([float(seconds)], dtype=self._torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

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

    def close(self):
        if self._collective and self._dist.is_initialized():
            self._dist.destroy_process_group()

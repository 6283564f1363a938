"""Two-stream (RGB + Flow) training: BASELINE configs[4].

The reference cannot express it (SURVEY.md 8d "Config 5": one modality per run, no fusion code), so this is the
shape-only synthetic the survey defines: two independent VideoModel-equivalents - one TrainEngine per feature stream, same
TA3N options - stepped on their own features, with the class logits SUMMED for reporting (late fusion).  Each stream is a
complete train step of the hot path (its own parameters, gradients, optimiser state and, under data parallelism, its own
all-reduce).  The two models are independent, so each steps on a HIP stream of its own: one model's launches fill the CUs and
the launch / prologue latencies the other leaves idle (measured at configs[4]'s shape: 470 us per two-stream step against 614
with both on one stream).  Every method returns with the caller's stream ordered behind both."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .engine import TrainEngine


class TwoStreamEngine:
    def __init__(self, batch_source: int, batch_target: int, num_segments: int, feature_dims: Sequence[int] = (1024, 1024),
                 fc_dim: int = 512, num_class: int = 12, concurrent: bool = True, **engine_kw):
        self.streams: List[TrainEngine] = [TrainEngine(batch_source, batch_target, num_segments, d, fc_dim, num_class, **engine_kw)
                                           for d in feature_dims]
        self.Bs, self.Bt, self.C = batch_source, batch_target, num_class
        dev = self.streams[0].device
        self._hip_streams = [torch.cuda.Stream(dev) for _ in self.streams] if concurrent else None

    def _each(self, fn) -> None:
        """fn(engine) for every model: on the model's own HIP stream, ordered behind the caller's stream on entry, and the
        caller's stream ordered behind it on exit."""
        if self._hip_streams is None:
            for e in self.streams:
                fn(e)
            return
        cur = torch.cuda.current_stream(self.streams[0].device)
        for e, s in zip(self.streams, self._hip_streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                fn(e)
        for s in self._hip_streams:
            cur.wait_stream(s)

    def set_batch(self, sources: Sequence[torch.Tensor], targets: Sequence[torch.Tensor], source_label: torch.Tensor) -> None:
        for e, xs, xt in zip(self.streams, sources, targets):
            e.set_batch(xs, xt, source_label)

    def train_step(self, beta, gamma, lr, pipelined: bool = True, **kw) -> None:
        self._each(lambda e: (e.train_step_pipelined if (pipelined and e.fused) else e.train_step)(beta, gamma, lr, **kw))

    def train_steps(self, schedule: Sequence[Sequence]) -> None:
        """len(schedule) two-stream steps enqueued by ONE call into the library (ta3n_train_steps_multi): step k of the RGB model on
        its HIP stream, step k of the Flow model on its own, step k + 1 of each, ... - no host call per step and stream, so a slow
        host core cannot stall either queue (round 3: the driver's box measured 1.46 ms per step with one Python call per step and
        stream against 0.5 ms of GPU work).  Bit-identical to train_step called once per entry."""
        if not schedule:
            return
        if not all(e.can_batch_steps() for e in self.streams):
            for beta, gamma, lr in schedule:
                self.train_step(beta, gamma, lr)
            return
        import ctypes as C
        from . import _lib
        dev = self.streams[0].device
        cur = torch.cuda.current_stream(dev)
        hip = self._hip_streams or [cur] * len(self.streams)
        jobs, keeps, runs = [], [], []
        for e, s in zip(self.streams, hip):
            if s is not cur:
                s.wait_stream(cur)
            with torch.cuda.stream(s):        # (the job records the stream that is current while it is built)
                job, keep, n_run = e._steps_job(schedule)
            jobs.append(job); keeps.append(keep); runs.append(n_run)
        n_run = runs[0]
        assert all(r == n_run for r in runs), "the models of a two-stream engine step together"
        if n_run:
            arr = (_lib.StepsJob * len(jobs))(*jobs)
            _lib.check(_lib.lib().ta3n_train_steps_multi(arr, len(jobs), n_run), "ta3n_train_steps_multi")
            for e, keep in zip(self.streams, keeps):
                e._steps_done(schedule, keep, n_run)
        for s in hip:
            if s is not cur:
                cur.wait_stream(s)

    def flush(self) -> None:
        self._each(lambda e: e.flush())

    def logits(self) -> torch.Tensor:
        """Summed class logits of the streams, [Bs + Bt, C] (what a two-stream evaluation reports)."""
        out = self.streams[0].outputs()["out"].clone()
        for e in self.streams[1:]:
            out += e.outputs()["out"]
        return out

    def losses(self) -> Dict[str, float]:
        tot: Dict[str, float] = {}
        for e in self.streams:
            for k, v in e.losses().items():
                tot[k] = tot.get(k, 0.0) + v
        return tot

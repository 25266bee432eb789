"""bench.py run end to end on a machine without a GPU — a dry run of its CONTROL FLOW, not a measurement: torch.cuda is replaced by
no-ops on CPU tensors, the liftover engine (not part of the host-side emulation) by a plan that hands back made-up records and kernel
times, and everything else is the real thing at toy size: the alignment generator, the column engine on the host-side emulation
(HGX_LIB_PATH = tests/cpp/build_cpu_emulation.sh's library), the oracle's CPU baselines and their parity gates.  The numbers it prints
mean nothing; that it prints ONE JSON line with every leg's keys is what tests/test_bench_dryrun.py checks."""
import sys
import time

import numpy as np
import torch


def install_leg_child(bench):
    """the child process of a leg (bench.run_leg_in_child) installs these fakes too: the dry run's own script says so — bench.py takes
    nothing of the kind from its environment"""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport bench_fakes; bench_fakes.install()\nimport bench\nbench.main()"
            % (bench.ROOT, here))
    bench.leg_child_command = lambda: [sys.executable, "-c", code]


def install():
    real_device = torch.device
    cpu = real_device("cpu")
    torch.device = lambda *a, **k: cpu

    class Stream:
        pass

    class Event:
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return max(1e-3, 1e3 * (other.t - self.t))
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.Stream = Stream
    torch.cuda.Event = Event
    torch.cuda.current_stream = lambda *a, **k: Stream()
    orig_to = torch.Tensor.to

    def to(self, *a, **k):  # (.to(dev) with the patched device: nothing to do)
        return orig_to(self, *a, **k)
    torch.Tensor.to = to

    import torch.distributed as dist
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **k: real_init("gloo")  # (N ranks on CPU tensors: gloo in the place of RCCL)

    import hal_amd

    class FakePlan:
        """hal_amd.LiftoverPlan's surface as bench.py uses it: three made-up records an interval"""
        def __init__(self, alignment, src, tgt, max_queries, traverse_dupes=True, min_length=0):
            self.nq = 0
            self.runs = 0
            self.timing = 1
            self._keep = None

        def _records(self, gs):
            n = int(gs.numel())
            self.nq = n
            rec = np.zeros(3 * n, dtype=hal_amd.RECORD_DTYPE)
            rec["query"] = np.repeat(np.arange(n), 3)
            rec["tgt_start"] = np.repeat(gs.numpy(), 3)
            rec["tgt_end"] = rec["tgt_start"] + 10
            rec["strand"] = b"+"
            self._keep = torch.from_numpy(rec.view(np.uint8).copy())
            return self._keep.data_ptr(), 3 * n

        def run(self, gs, ge, st):
            self.runs += 1
            return self._records(gs)

        def submit(self, gs, ge, st, stream=None):
            self._pending = self._records(gs)

        def collect(self):
            self.runs += 1
            return self._pending

        def set_timing(self, mode):
            self.timing = mode
            if mode == 2:
                self.runs = 0

        def set_workers(self, n):
            pass

        def stats(self):
            return dict(queries=self.nq, records=3 * self.nq, general_queries=1, deferred_queries=0, composed_kind=3, composed_records=1000,
                        composed_build_ms=1.0, top_derefs=10 * self.nq, bottom_derefs=10 * self.nq, source_pieces=2 * self.nq, mapped_pieces=4 * self.nq)

        def kernel_times(self):
            k = max(1, self.runs)
            return {"k_lift_classify": {"ms": 0.05 * k, "launches": k, "top_derefs": 5 * k, "bot_derefs": 0},
                    "k_lift_totals": {"ms": 0.007 * k, "launches": k, "top_derefs": 0, "bot_derefs": 0},
                    "k_lift_merged": {"ms": 0.045 * k, "launches": k, "top_derefs": 3 * self.nq * k, "bot_derefs": 0},
                    "k_up_chain": {"ms": 1.0 * k, "launches": k, "top_derefs": 10 * self.nq * k, "bot_derefs": 10 * self.nq * k}}

        def records_to_tensor(self, ptr, n, packed=False):
            return self._keep[:n * 40]

        def wire_capacity(self):
            return 32 + 2 * self.nq + 8 + 12 * 3 * self.nq

        def wire_blob(self, first_query=0, dst=None, bed_only=False):
            from hal_amd import shard
            blob = shard.encode_blob(self._keep.view(-1, 40), self.nq, first_query=first_query, fmt=8 if bed_only else None)
            if dst is not None:
                dst[:blob.numel()] = blob
                blob = dst[:blob.numel()]
            return blob, int(blob[4])
    hal_amd.LiftoverPlan = FakePlan
    hal_amd.liftover_convert_bytes = lambda al, src, data, tgt, **k: (3 * len(data), 3 * data.count(b"\n"))
    hal_amd.build_phases = lambda: [("a phase", 0.1)]
    hal_amd.Alignment.blocks_in_target_ranges = lambda self, q, t, chrom, ranges, **k: [([], []) for _ in ranges]
    hal_amd.format_block_results = lambda blocks, dupes: ""
    hal_amd.maf_export_multi = lambda clones, ref, *a, **k: hal_amd.Alignment.maf_export_bytes(clones[0], ref, start=k.get("start", 0), length=k.get("length", 0),
                                                                                                 no_ancestors=k.get("no_ancestors", False), unique=k.get("unique", False))

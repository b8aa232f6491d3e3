"""Can an admission prefill and the decode ticks of live rows share the GPU?  One thread replays the captured decode step of an 8-row batcher
(read-back per tick, as ContinuousBatcher.step does), another runs the admission prefill of k images on a stream of its own; each alone, then
both.   python tests/diag/overlap_probe.py [rows=8] [k=4]"""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import config, constants, engine, ops, synth
from groma_amd.groma import GromaModel
from groma_amd.serving import ContinuousBatcher, _RowView

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = config.groma_7b(box_score_thres=0.0)
m = GromaModel.from_synthetic(cfg, seed=0, device="cuda", precision="bf16")
m.init_special_token_id(constants.SyntheticTokenizer())
m.generation_config.eos_token_id = None
images, ids = synth.make_inputs(cfg, m, k, seed=5)
images = images.cuda()
b = ContinuousBatcher(m, max_rows=rows, max_len=1024)
b.step()
b.warm_admission(ids[0], images[0], rows=k)
idsk = torch.stack([ids[i] for i in range(k)]).cpu()


def prefill():
    with ops.precision(m.precision if isinstance(m.precision, str) and m.precision in ("bf16", "fp16", "ref") else "bf16"):
        m.forward(input_ids=idsk.clone(), images=images, use_cache=True, return_dict=True, _cache=_RowView(b.staging, k), _seeds=list(range(k)))


HI = torch.cuda.Stream(priority=-1)   # the ticks' stream in the "priority" runs: its kernels are dispatched ahead of the prefill's


def ticks(stop, out, hi=False):
    n = 0
    t = time.perf_counter()
    with torch.cuda.stream(HI if hi else torch.cuda.current_stream()):
        while not stop.is_set():
            b.graph.replay()
            b.tok.tolist()
            n += 1
    out.append((n, time.perf_counter() - t))


for _ in range(3):
    prefill()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    prefill()
torch.cuda.synchronize()
p_alone = (time.perf_counter() - t) / 5
t = time.perf_counter()
for _ in range(50):
    b.graph.replay(); b.tok.tolist()
d_alone = (time.perf_counter() - t) / 50
print(f"alone: prefill of {k} images {p_alone * 1e3:.1f} ms; decode tick ({rows} rows) {d_alone * 1e3:.2f} ms", flush=True)

for yld, hi in ((False, False), (True, False), (False, True), (True, True)):
    stop, out = threading.Event(), []
    side = torch.cuda.Stream()
    NP = 6

    def worker():
        with torch.cuda.stream(side), ops.gemm_yield(yld):
            for _ in range(NP):
                prefill()
            side.synchronize()
        stop.set()

    th = threading.Thread(target=worker)
    torch.cuda.synchronize()
    t = time.perf_counter()
    th.start()
    ticks(stop, out, hi)
    th.join()
    dt = time.perf_counter() - t
    n, _ = out[0]
    # time the same work would take back to back
    serial = NP * p_alone + n * d_alone
    print(f"together (gemm_yield={yld}, ticks on a high-priority stream={hi}): {NP} prefills + {n} ticks in {dt * 1e3:.0f} ms  (prefill {dt / NP * 1e3:.1f} ms each, tick {dt / max(n, 1) * 1e3:.2f} ms); "
          f"the same work back to back {serial * 1e3:.0f} ms -> x{serial / dt:.2f}", flush=True)

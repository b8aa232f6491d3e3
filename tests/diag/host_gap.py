"""Where the host spends the NMS round trip of a forward (GPU idle meanwhile): wall-clock segments of GromaModel.propose() and of the
splice / upload section of forward(), Groma-7B, 14 images.  usage: python tests/diag/host_gap.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import config as gconfig, constants, synth
from groma_amd.groma import GromaModel

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 14
cfg = gconfig.groma_7b()
m = GromaModel.from_synthetic(cfg, seed=0, device=torch.device("cuda"))
m.init_special_token_id(constants.SyntheticTokenizer())
images, ids = synth.make_inputs(cfg, m, bs, seed=1, prompt_len=128)
images, ids = images.cuda(), ids.cuda()
for _ in range(4):
    m.forward(input_ids=ids.clone(), images=images, return_dict=True)
torch.cuda.synchronize()

# time individual host operations of the round trip on the live tensors of the last forward
aux = m._last_aux
keep = torch.stack([torch.nn.functional.pad(k, (0, 100 - k.numel())) for k in aux["nms_keep"]]).cuda()
n_keep = torch.tensor([k.numel() for k in aux["nms_keep"]], dtype=torch.int32).cuda()
torch.cuda.synchronize()


def t(label, fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    print(f"{label:60s} {dt:8.1f} us")
    return r


kk_h = t("cat + .cpu() (pageable D2H, GPU idle)", lambda: torch.cat([keep, n_keep.to(torch.int64)[:, None]], dim=1).cpu())
pin = torch.empty((bs, 101), dtype=torch.int64).pin_memory()
def pinned():
    pin.copy_(torch.cat([keep, n_keep.to(torch.int64)[:, None]], dim=1), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return pin
t("cat + pinned async D2H + stream sync", pinned)
t("14 x (randperm + index_select)", lambda: [kk_h[i, :100].index_select(0, torch.randperm(100)) for i in range(bs)])
sel = [kk_h[i, :100].index_select(0, torch.randperm(100)) for i in range(bs)]
def stage():
    n_sel = [int(x.numel()) for x in sel]
    R = sum(n_sel)
    st = m._pinned("sel", 2 * R)
    img_of = torch.repeat_interleave(torch.arange(bs), torch.tensor(n_sel))
    torch.add(torch.cat(sel), img_of, alpha=300, out=st[:R])
    st[R:2 * R] = img_of
    return st[:2 * R].to("cuda", non_blocking=True)
t("stage + H2D launch", stage)
ids_h = ids.cpu()
n_reg = [100] * bs
t("_splice (host)", lambda: m._splice(ids_h, 256, n_reg))
new_ids_h, mask_h = m._splice(ids_h, 256, n_reg)
def rows():
    flat = new_ids_h.reshape(-1)
    a = (flat == m.img_token_id).nonzero(as_tuple=True)[0]
    b = (flat == m.reg_token_id).nonzero(as_tuple=True)[0]
    c = (flat == m.refer_feat_token_id).nonzero(as_tuple=True)[0]
    return a, b, c, mask_h.sum(-1).tolist(), bool(mask_h.all())
t("row lists + mask bookkeeping (host)", rows)
t("ids.cpu() of the caller's device input_ids", lambda: ids.cpu())
# whole forward, wall clock
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    m.forward(input_ids=ids.clone(), images=images, return_dict=True)
torch.cuda.synchronize()
print(f"forward: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call at {bs} images")

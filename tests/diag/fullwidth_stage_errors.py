"""GPU diag: where the full-width device path departs from the bf16-rounded oracle, sub-stage by sub-stage."""
import os, sys, math
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from groma_amd import config as gconfig, synth, ops
from oracle import groma_oracle as O
from tests import util

torch.set_num_threads(64)
cfg = gconfig.groma_7b_width(box_score_thres=0.0)
sd = synth.make_state_dict(cfg, 0)
tk = util.TokenIds()
model = util.device_model(cfg, sd)
cd = cfg.to_dict()
images, ids = synth.make_inputs(cfg, tk, bs=1, seed=1234)
rel = util.relerr
with torch.no_grad():
    h4 = [h.clone() for h in model.vit.forward(images.cuda())]
    hc = [h.cpu() for h in h4]
    with O.rounding("bf16"):
        ref_v = O.vit_forward(sd, cd, images)[-4:]
    print("vit vs bf16 oracle", [rel(a, b) for a, b in zip(hc, ref_v)])
    # ---- region fuse
    feats, S = model.region.fuse(h4[-3:])
    feats_c = [f.float().cpu().permute(0, 3, 1, 2) for f in feats]
    mlvl = [h[:, 1:] for h in hc[-3:]]
    for mode in (None, "bf16"):
        with O.rounding(mode):
            rf = O.region_fuse(sd, cd, mlvl)
        print("fuse feats vs", mode, [rel(a, b) for a, b in zip(feats_c, rf)])
    # input conv outputs (device workspace) vs oracle
    g = 32
    with O.rounding("bf16"):
        fe = [t.reshape(1, g, g, 1024).permute(0, 3, 1, 2) for t in mlvl]
        to_shape = [(128, 128), (64, 64), (32, 32)]
        fe = [F.interpolate(f, size=s, mode="bilinear", align_corners=True) for f, s in zip(fe, to_shape)]
        for lvl, f in enumerate(fe):
            H, W = f.shape[-2:]
            y, x = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
            f2 = torch.cat([f, x.expand(1, 1, -1, -1), y.expand(1, 1, -1, -1)], 1)
            ref_in = O._r(O._conv16(f2, sd[f"region_encoder.mlvl_fuse.input_conv.{lvl}.weight"], sd[f"region_encoder.mlvl_fuse.input_conv.{lvl}.bias"]))
            dev_in = model._ws.get(f"reg_in{lvl}", (H * W, 1024), torch.bfloat16).float().cpu().view(1, H, W, 1024).permute(0, 3, 1, 2)
            up = ops.upsample_coord_pack(h4[-3 + lvl], 32, H, model.region.w["Cpad"]).float().cpu().view(1, H, W, -1).permute(0, 3, 1, 2)
            print(f"level {lvl}: upsample+coord vs oracle {rel(up[:, :1026], O._r(f2)):.3e}  input conv {rel(dev_in, ref_in):.3e}")
    # ---- extract on ORACLE-rounded feats (fed to both)
    with O.rounding("bf16"):
        rf = O.region_fuse(sd, cd, mlvl)
    torch.manual_seed(0)
    boxes = torch.rand(100, 4) * torch.tensor([1.0, 1.0, 0.5, 0.5])
    dev_feats = [f.permute(0, 2, 3, 1).contiguous().bfloat16().cuda() for f in rf]
    out = model.region.extract(dev_feats, S, boxes.cuda(), torch.zeros(100).cuda()).cpu()
    for mode in (None, "bf16"):
        with O.rounding(mode):
            r = torch.cat(O.roi_extract(sd, cd, rf, [boxes]))
        print("extract (same feats) vs", mode, rel(out, r))
    # pieces of extract: pconv output
    P, D, R = 14, 1024, 100
    pc = model._ws.get("reg_pc", (R * P * P, D), torch.bfloat16).float().cpu().view(R, P, P, D).permute(0, 3, 1, 2)
    from oracle import cref
    with O.rounding("bf16"):
        rois = torch.cat([torch.zeros(100, 1), boxes * 448], 1)
        acc = None
        for lvl, f in enumerate(rf):
            rfe = torch.from_numpy(cref.roi_align_avg(f.float().contiguous().numpy(), rois.numpy(), (14, 14), 1.0 / [14 / 8, 14 / 4, 14 / 2][lvl], 2, True))
            tiles = model._ws.get("reg_tiles", (3, R, P + 2, P + 2, D), torch.bfloat16)[lvl, :, 1:-1, 1:-1].float().cpu().permute(0, 3, 1, 2)
            print(f"  roialign level {lvl} tiles vs rounded oracle {rel(tiles, O._r(rfe)):.3e}")
            y = O._conv16(rfe, sd[f"region_encoder.roi_align.pconvs.{lvl}.weight"], sd[f"region_encoder.roi_align.pconvs.{lvl}.bias"], padding=1)
            acc = y if acc is None else acc + y
        print("  pconv(relu) vs rounded oracle", rel(pc, O._r(F.relu(acc))))
        x = O._lin16(F.relu(acc).flatten(1, -1), sd, "region_encoder.roi_align.flatten_linear")
        print("  flatten out norm", x.norm().item())
    # ---- LLaMA on the device's own embeds
    model.capture_embeds = True
    torch.manual_seed(77)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True)
    emb = model._last_aux["inputs_embeds"].cpu()
    L = emb.shape[1]
    for mode in (None, "bf16"):
        with O.rounding(mode):
            hid, past = O.llama_forward(sd, cd, emb, torch.ones(1, L))
            lg = O.lm_logits(sd, hid)
        print("llama logits vs", mode, rel(out.logits.float().cpu(), lg), "K", rel(out.past_key_values[0][0].float().cpu(), past[0][0]),
              "V", rel(out.past_key_values[0][1].float().cpu(), past[0][1]))

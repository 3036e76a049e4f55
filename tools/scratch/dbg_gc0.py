import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from hicom_amd import native as nv
dev = torch.device("cuda", 0)
m = bench.make_projector(bench.release_config(896, 64), dev)
gc = m.global_compressor
c = gc.readout_over_out_proj()
want = gc.readout[0].weight.float() @ gc.attn_layer.out_proj.weight.float()
torch.cuda.synchronize()
print("gc0", c.dtype, tuple(c.shape), "max err vs fp32 product", float((c.float() - want).abs().max()), "max |C|", float(want.abs().max()))
# the aux role alone: GELU(C x + r0)
E, HID = 1152, 896
po = torch.randn(18, E, device=dev); bv = torch.zeros(E, device=dev).bfloat16(); r0 = torch.randn(HID, device=dev)
a1 = nv.to_f16(torch.randn(96, 64, device=dev)); w1 = nv.to_f16(torch.randn(64, 64, device=dev)); o1 = torch.empty(96, 64, device=dev, dtype=torch.float16)
hid = torch.empty(HID, device=dev)
for wmat in (c, want.contiguous()):
    nv.readout16_gemm(a1, w1, None, out_f16=o1, aux=dict(xs=po, xb=bv, w=wmat, b=r0, act=1, y=hid))
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(want @ po.sum(0) + r0)
    print(wmat.dtype, "aux max err", float((hid - ref).abs().max()))

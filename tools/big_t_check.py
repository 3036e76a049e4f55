import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
for T in (128, 256, 512):
    m = bench.make_projector(bench.release_config(896, T), dev)
    g = torch.Generator(device=dev).manual_seed(1)
    ff = torch.randn(T, 27, 27, 1152, device=dev, generator=g).bfloat16()
    fe = torch.randn(T, 27, 27, 1152, device=dev, generator=g).bfloat16()
    gd = torch.randn(1152, device=dev, generator=g).bfloat16()
    with torch.no_grad():
        out = m(ff, fe, gd, "video", None)
        torch.cuda.synchronize()
        m64 = bench.make_projector(bench.release_config(896, T), dev)
        o64 = m64(ff[:64].contiguous(), fe[:64].contiguous(), gd, "video", None)
        torch.cuda.synchronize()
    nloc = 64 // 4 * 81
    print(T, tuple(out.shape), bool(torch.isfinite(out.float()).all()), "local rows of the first 64 frames vs a 64-frame call:", float((out[:nloc].float() - o64[:nloc].float()).abs().max()))
    del ff, fe, out, m, m64
    torch.cuda.empty_cache()

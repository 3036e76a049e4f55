"""Dev tool: host time of each statement of sharded_forward (world 1, RCCL), accumulated over short bursts."""
import os, socket, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import torch.distributed as dist
from hicom_amd import dist as hd, native as nv
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
cfg = bench.release_config(896, 64); m = bench.make_projector(cfg, dev)
ff = torch.randn(64, 27, 27, 1152, device=dev).bfloat16(); fe = torch.randn_like(ff); g = torch.randn(1152, device=dev).bfloat16()
acc = {}
def T(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
def step():
    t = time.perf_counter()
    plan = hd._shard_plan(m, ff, fe, g, 64, None, None); t = T("plan lookup", t)
    st = plan.sets[plan.xs.advance()]
    main, comm = torch.cuda.current_stream(dev), plan.comm; t = T("current_stream", t)
    out = torch.empty((plan.n_rows_total, plan.hidden), dtype=plan.odt, device=dev); t = T("empty", t)
    out.record_stream(comm); t = T("record_stream", t)
    main.wait_event(st.ev_tok); t = T("wait ev_tok", t)
    st.a_stream.out = st.a_finish.out = out.data_ptr()
    nv.compressor_fwd(st.a_stream); t = T("C stream phase", t)
    st.ev_stream.record(main); t = T("record ev_stream", t)
    with torch.cuda.stream(comm):
        t = T("enter stream ctx", t)
        comm.wait_event(st.ev_stream); t = T("wait ev_stream", t)
        dist.all_gather_into_tensor(st.everyone.view(-1), st.mine); t = T("all_gather", t)
        nv.compressor_fwd(st.a_finish); t = T("C finish phase", t)
        nv.place_blocks(st.everyone.data_ptr() + st.tok_off, plan.nw, plan.world, st.mine.numel(), plan.hidden * 2, out, 0,
                        nl_group=plan.lay.nl_group, stream=comm.cuda_stream); t = T("place_blocks", t)
        st.ev_tok.record(comm); t = T("record ev_tok", t)
    t = T("exit stream ctx", t)
    return out
with torch.no_grad():
    for _ in range(10): hd.sharded_forward(m, ff, fe, g, 64, deferred=True)
    torch.cuda.synchronize()
    n = 0
    for rep in range(20):
        torch.cuda.synchronize()
        for _ in range(6): step(); n += 1
    torch.cuda.synchronize()
tot = 0
for k, v in acc.items():
    print("%-20s %6.1f us" % (k, v / n * 1e6)); tot += v
print("%-20s %6.1f us" % ("total", tot / n * 1e6))
dist.destroy_process_group()

#!/usr/bin/env python3
"""RGB_DEBUG=16 on the profiling build (make -C ra_amd/csrc prof): per-wave timestamps of one class-dispatch
tick -> concurrency picture."""
import os, sys, ctypes as C
os.environ["RGB_DEBUG"] = os.environ.get("TL_DBG", "16")
os.environ.setdefault("RGB_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "ra_amd", "csrc", "libra_gpu_batch_prof.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ra_amd import abi, engine, workload as W
CFG = os.environ.get("TL_CONFIG")                     # "5" / "3": the literal SURVEY 8(d) configuration (bench.LITERAL)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); sp = stream.cuda_stream
if CFG:
    import bench
    from oracle import oracle as O
    c = bench.LITERAL[CFG]
    G, N, seed = c["groups"], c["members"], c["seed"]
    S = G * N
    st0 = W.initial_states(G, N, seed, **c["init"])
    cpu = O.Oracle(G, N, max_runs=16); cpu.set_state(0, st0)
    ticks = []
    for t in range(int(os.environ.get("TL_TICKS", "8"))):
        m = W.gen_tick(cpu.get_state(), N, t, seed, getattr(W, c["mix"]), **c["gen"])
        cpu.step_parallel(m); ticks.append(m)
    cpu.close()
    eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
    eng.set_state(0, st0)
    stride = max(len(m) for m in ticks)
    dd = torch.empty(stride * 64, dtype=torch.uint8, device="cuda")
    dr = torch.empty(stride * max(N - 1, 1) * 56, dtype=torch.uint8, device="cuda")
    dn = torch.tensor([len(ticks[-1])], dtype=torch.int32, device="cuda")
    kc = torch.from_numpy(np.bincount(ticks[-1]["kind"], minlength=abi.N_KINDS)[:abi.N_KINDS].astype(np.int32)).cuda()
    for m in ticks:
        dm = torch.from_numpy(m.view(np.uint8)).cuda()
        k1 = np.bincount(m["kind"], minlength=abi.N_KINDS)[:abi.N_KINDS].astype(np.uint32).reshape(1, -1)
        eng.run_ticks_device(dm.data_ptr(), len(m), 1, dd.data_ptr(), dr.data_ptr(), sp,
                             tick_counts=np.array([len(m)], dtype=np.uint32), kind_counts=k1)
        torch.cuda.synchronize()
else:
    G, N = int(os.environ.get('TL_GROUPS', '65536')), 5
    S = G * N
    eng = engine.RaGpuBatch(G, N, max_runs=16, ring_slots=1, ring_capacity=64)
    eng.set_state(0, W.initial_states(G, N, 0x5EED0003))
    dm = torch.empty(S * 64, dtype=torch.uint8, device="cuda"); dd = torch.empty(S * 64, dtype=torch.uint8, device="cuda")
    dr = torch.empty(S * 4 * 56, dtype=torch.uint8, device="cuda")
    kc = torch.zeros(abi.N_KINDS, dtype=torch.int32, device="cuda"); dn = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(int(os.environ.get("TL_TICKS", "24"))):
        eng.synth_tick_device(0x5EED0003, t, dm.data_ptr(), kc.data_ptr(), dn.data_ptr(), sp)
        torch.cuda.synchronize()
        eng.synth_apply_tick_device(dm.data_ptr(), S, dd.data_ptr(), dr.data_ptr(), sp)
        torch.cuda.synchronize()
nblk = S // 64 + 16
buf = np.zeros(nblk * 8, dtype=np.uint64)
L = engine.lib(); L.rgb_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
assert L.rgb_debug_read(eng._h, buf.ctypes.data, len(buf)) == 0
b = buf.reshape(nblk, 8)
b = b[b[:, 0] > 0]
b = b[b[:, 0].astype(np.int64) > b[:, 0].astype(np.int64).max() - 5000]   # the last launch only (50 us window)
cls = (b[:, 2] >> np.uint64(60)).astype(int)
M40 = np.uint64((1 << 40) - 1)
t0 = b[:, 0].astype(np.int64); t1 = (b[:, 1] & M40).astype(np.int64); dl = (b[:, 1] >> np.uint64(40)).astype(np.int64)
t2 = (b[:, 2] & M40).astype(np.int64); dst = ((b[:, 2] >> np.uint64(40)) & np.uint64(0xFFFFF)).astype(np.int64); t3 = (b[:, 3] & M40).astype(np.int64)
# wavefronts whose lanes all took a fast path never ran the general clause code: t2 (its end) is unset.  Their
# "process" phase ends where the fast paths ended (o[7] >> 24, relative to the messages-in-registers stamp) and
# their state-load figure (stamped inside the general path) does not exist
tfast_all = (b[:, 7] >> np.uint64(24)).astype(np.int64)
only_fast = t2 == 0
t2 = np.where(only_fast, (t1 & ((1 << 40) - 1)) + tfast_all, t2)
dl = np.where(only_fast, 0, dl)
t0 = t0 & ((1 << 40) - 1)
z = t0.min()
tick = 10.0  # ns per wall_clock64 tick (100 MHz)
print("waves", len(b), "msgs", int(dn.item()), "kernel span %.1f us" % ((t3.max() - z) * tick / 1e3))
for c in range(12):
    m = cls == c
    if not m.any(): continue
    print(f"class {c}: waves {m.sum():5d} start {((t0[m]-z).min()*tick/1e3):6.1f}..{((t0[m]-z).max()*tick/1e3):6.1f} us | "
          f"msg-load {np.median(t1[m]-t0[m])*tick/1e3:5.2f} | state-load {(f'{np.median(dl[m][dl[m] > 0])*tick/1e3:5.2f}' if (dl[m] > 0).any() else '  n/a')} ({int((dl[m] > 0).sum())} of {int(m.sum())} waves ran the general path) | store-drain {np.median(dst[m])*tick/1e3:5.2f} | process {np.median(t2[m]-t1[m])*tick/1e3:5.2f} (p90 {np.percentile(t2[m]-t1[m],90)*tick/1e3:5.2f}) | "
          f"dec-store {np.median(t3[m]-t2[m])*tick/1e3:5.2f} | wave life {np.median(t3[m]-t0[m])*tick/1e3:5.2f} us")
    M24 = (1 << 24) - 1
    d_disp, d_comm = (b[m, 5] >> np.uint64(24)).astype(np.int64), (b[m, 6] >> np.uint64(24)).astype(np.int64)
    ok = d_disp > 0
    if ok.any():
        print(f"          lane 0, from messages-in-registers: state loaded {np.median(dl[m][ok])*tick/1e3:5.2f} | clause code done {np.median(d_disp[ok])*tick/1e3:5.2f} | commit issued {np.median(d_comm[ok])*tick/1e3:5.2f} | process returned {np.median((t2[m]-t1[m])[ok])*tick/1e3:5.2f} us")
    mx, sm, nz, cn = b[m, 4].astype(int), (b[m, 5] & np.uint64(M24)).astype(int), (b[m, 6] & np.uint64(M24)).astype(int), (b[m, 7] & np.uint64(0xFF)).astype(int)
    nfast, tfast = ((b[m, 7] >> np.uint64(8)) & np.uint64(0xFF)).astype(int), (b[m, 7] >> np.uint64(24)).astype(np.int64)
    if c <= 2 and (tfast > 0).any():
        gen = nfast < cn                 # wavefronts that went on to the general clause code
        print(f"          fast paths done {np.median(tfast[tfast > 0])*tick/1e3:5.2f} us after messages-in-registers; lanes they took {nfast.sum()}/{cn.sum()} "
              f"({100.0*nfast.sum()/max(cn.sum(),1):.1f} %); wavefronts left with general-path lanes {gen.sum()}/{len(cn)}, "
              f"wave life with / without: {np.median((t3[m]-t0[m])[gen])*tick/1e3 if gen.any() else float('nan'):.1f} / "
              f"{np.median((t3[m]-t0[m])[~gen])*tick/1e3 if (~gen).any() else float('nan'):.1f} us")
    print(f"          run-table words read: lanes reading {nz.sum()}/{cn.sum()} ({100.0*nz.sum()/max(cn.sum(),1):.1f} %), per reading lane {sm.sum()/max(nz.sum(),1):.1f}, "
          f"waves with a reader {100.0*(nz>0).mean():.0f} %, wave max p50/p90/max {np.percentile(mx,50):.0f}/{np.percentile(mx,90):.0f}/{mx.max()}; "
          f"wave life by wave-max words 0/1-2/3-6/7+: " + "/".join(
              f"{np.median((t3[m]-t0[m])[sel])*tick/1e3:.1f}" if sel.any() else "-" for sel in (mx == 0, (mx >= 1) & (mx <= 2), (mx >= 3) & (mx <= 6), mx >= 7)))
# who is the tail: end times per class, and the last waves to finish
endt = (t3 - z) * tick / 1e3; startt = (t0 - z) * tick / 1e3
print("end time per class (us): class waves p50 p90 p99 max | waves ending in the last 4 us of the kernel")
for c in range(15):
    m = cls == c
    if not m.any(): continue
    e = endt[m]
    print(f"  class {c:2d} {m.sum():5d}  {np.percentile(e,50):5.1f} {np.percentile(e,90):5.1f} {np.percentile(e,99):5.1f} {e.max():5.1f} | {(e > endt.max() - 4).sum()}")
order = np.argsort(-endt)[:24]
print("last waves: class start end life lanes max-run-words")
for i in order:
    print(f"  {cls[i]:2d} {startt[i]:5.1f} {endt[i]:5.1f} {endt[i]-startt[i]:5.1f} {int(b[i,7]) & 0xFF:3d} {int(b[i,4]):3d}")
# concurrency over time
ev = np.concatenate([np.stack([t0 - z, np.ones_like(t0)], 1), np.stack([t3 - z, -np.ones_like(t3)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
for us in range(0, min(int((t3.max() - z) * tick / 1e3) + 1, 60), 2):
    i = np.searchsorted(ev[:, 0], us * 1e3 / tick)
    print(f"  t={us:3d} us  waves in flight {int(conc[min(i, len(conc)-1)]):5d}  started {int((t0 - z <= us*1e3/tick).sum()):5d}")
st = eng.get_state()
import collections
print("n_runs hist", np.bincount(st["n_runs"], minlength=17).tolist())
print("roles", np.bincount(st["role"], minlength=5).tolist())
print("log span (LI-first) p50/p99", np.percentile((st["last_index"] - st["first_index"]).astype(np.int64), [50, 99]).tolist())
print("LI-LWI p50/p99", np.percentile((st["last_index"] - st["last_written_index"]).astype(np.int64), [50, 99]).tolist())
print("kind counts", kc.cpu().numpy().tolist())

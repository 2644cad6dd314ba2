import sys, os, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import miniasm_amd as ma, refapi as R, stages as ST
paf = R.pafgen('/tmp/dbg.paf', 1500, 40000, 7, ["-L", "uniform", "-d", "0.3", "-x", "0.03"])
opt = ma.default_opt()
ing = ma.Ingest(paf, opt)
ctx = ma.Ctx(0)
O = ST.orc_stages(ing.hits, ing.n_seq, opt)
ns = O["n_seq_new"]
ctx.hits_upload(ing.hits, ing.n_seq); ctx.sort()
ctx.sub(opt.min_dp, opt.min_iden, 0, 0); ctx.cut(0, opt.min_span); ctx.flt(0, *ST.flt_params(opt))
ctx.sub(opt.min_dp, opt.min_iden, opt.min_span // 2, 1); ctx.cut(1, opt.min_span); ctx.sub_merge(); ctx.contained(opt)
ctx.sg_gen(opt, True)
sg, seq, idx = ctx.asg_download()
print("sg equal:", sg.tobytes() == O["sg_arcs"].tobytes(), len(sg))
# oracle marking only
arcs = O["sg_arcs"].copy(); oidx = np.zeros(2*ns, dtype='<u8')
R.orc().orc_arc_index(ns, len(arcs), arcs.ctypes.data, oidx.ctypes.data)
sdel = (O["sg_seq"] >> 31).astype(np.uint8)
inner = C.c_uint64(0)
nr = R.orc().orc_arc_del_trans(ns, len(arcs), arcs.ctypes.data, oidx.ctypes.data, sdel.ctypes.data, opt.gap_fuzz, C.byref(inner))
# gpu marking + cleanup
g_nr = ctx.del_trans(opt.gap_fuzz)
ga, _, _ = ctx.asg_download()
print("n_red oracle", nr, "gpu", g_nr, "idx equal", idx.tobytes() == oidx.tobytes())
keep = arcs[(arcs["oldel"] >> 31) == 0]
print("after trans+rm: oracle", len(keep), "gpu", len(ga))
so = set((int(a["ul"]), int(a["v"])) for a in keep); sg_ = set((int(a["ul"]), int(a["v"])) for a in ga)
miss = sorted(so - sg_); extra = sorted(sg_ - so)
print("oracle-only", len(miss), "gpu-only", len(extra))
for (ul, v) in miss[:6]:
    u = ul >> 32
    st, n = int(oidx[u]) >> 32, int(oidx[u]) & 0xffffffff
    print("arc u=%d v=%d len=%d ; u list (v,len):" % (u, v, ul & 0xffffffff), [(int(a["v"]), int(a["ul"]) & 0xffffffff) for a in O["sg_arcs"][st:st+n]])
if g_nr:
    print("symm gpu", ctx.symm())
print("oracle cnt", O["tr_cnt"])

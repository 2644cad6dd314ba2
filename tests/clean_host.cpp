// clean_host.cpp -- TEST HARNESS, not product code: runs the per-vertex device functions of the graph cleaners
// (miniasm_amd/csrc/clean_core.h) and of the unitig construction (miniasm_amd/csrc/ug_core.h) on the CPU, one "launch" =
// one loop over the vertices, so that the fixpoint / pointer-jumping algorithms can be pinned against the reference library
// without a GPU (tests/test_clean_core_cpu.py).  The HIP kernels in csrc/clean.hip and csrc/ug.hip call the very same functions.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "clean_core.h"
#include "ug_core.h"

struct arc_t { uint32_t len, u, v, ol; }; // bytes of asg_arc_t (asg.h:7-11): ul = u<<32|len, v, ol:31|del:1

struct Soa {
	std::vector<uint32_t> au, av, alen, aol, slen;
	std::vector<uint8_t> sdel;
	std::vector<unsigned long long> idx;
};

static void to_soa(uint32_t n_seq, uint32_t n_arc, const arc_t *arc, const uint64_t *idx, const uint32_t *seq, Soa &g)
{
	g.au.resize(n_arc + 1); g.av.resize(n_arc + 1); g.alen.resize(n_arc + 1); g.aol.resize(n_arc + 1);
	for (uint32_t i = 0; i < n_arc; ++i) g.au[i] = arc[i].u, g.av[i] = arc[i].v, g.alen[i] = arc[i].len, g.aol[i] = arc[i].ol;
	g.slen.resize(n_seq + 1); g.sdel.resize(n_seq + 1);
	for (uint32_t r = 0; r < n_seq; ++r) g.slen[r] = seq[r] & 0x7fffffffu, g.sdel[r] = (uint8_t)(seq[r] >> 31);
	g.idx.assign(idx, idx + 2 * (size_t)n_seq);
	g.idx.push_back(0);
}

// asg_pop_bubble as the reference runs it (clean_core.h: cl_bubble_sweep_seq; csrc/clean.hip: k_clean_bubble_seq): sets / clears the del bits in arc[] / seq[] (no cleanup)
extern "C" int clh_bubble_seq(int max_dist, uint32_t n_seq, uint32_t n_arc, arc_t *arc, const uint64_t *idx, uint32_t *seq, uint32_t *cnt, uint32_t *cnt2)
{
	Soa G;
	to_soa(n_seq, n_arc, arc, idx, seq, G);
	const uint32_t V = 2 * n_seq;
	std::vector<cl_seqinfo_t> info(V + 1);
	memset(info.data(), 0, info.size() * sizeof(cl_seqinfo_t));
	std::vector<uint32_t> stk(V + 1), seen(V + 1), walked(n_arc + 1);
	unsigned long long pops = 0, tips = 0;
	if (cl_bubble_sweep_seq(G.au.data(), G.av.data(), G.alen.data(), G.aol.data(), G.idx.data(), G.sdel.data(), V, (uint32_t)max_dist, info.data(), stk.data(), seen.data(), walked.data(), &pops, &tips) != 0) return -1;
	*cnt = (uint32_t)pops; *cnt2 = (uint32_t)tips;
	for (uint32_t r = 0; r < n_seq; ++r) seq[r] = (seq[r] & 0x7fffffffu) | (uint32_t)G.sdel[r] << 31;
	for (uint32_t e = 0; e < n_arc; ++e) arc[e].ol = G.aol[e];
	return 0;
}

// mode 0 tip, 1 internal, 2 bi-loop (param = max_ext), 3 bubbles (param = max_dist).  Sets the del bits in arc[] / seq[] (no cleanup).
extern "C" int clh_sweep(int mode, int param, uint32_t n_seq, uint32_t n_arc, arc_t *arc, const uint64_t *idx, uint32_t *seq,
                         uint32_t *cnt, uint32_t *cnt2, int *iters, uint32_t bubble_cap)
{
	Soa G;
	to_soa(n_seq, n_arc, arc, idx, seq, G);
	const uint32_t V = 2 * n_seq;
	std::vector<uint32_t> rst[2], ast[2];
	for (int k = 0; k < 2; ++k) rst[k].assign(n_seq + 1, CL_NONE), ast[k].assign(n_arc + 1, CL_NONE);
	cl_view_t g;
	g.av = G.av.data(); g.alen = G.alen.data(); g.aol = G.aol.data(); g.idx = G.idx.data(); g.sdel = G.sdel.data(); g.n_vtx = V; g.no_stamps = 0;
	uint32_t cap = bubble_cap ? bubble_cap : 64;
	int cur = 0;
	*cnt = *cnt2 = 0; *iters = 0;
	for (int it = 0; it < 100000; ++it) {
		std::fill(rst[cur ^ 1].begin(), rst[cur ^ 1].end(), CL_NONE);
		std::fill(ast[cur ^ 1].begin(), ast[cur ^ 1].end(), CL_NONE);
		cl_stamps_t s; s.rst = rst[cur ^ 1].data(); s.ast = ast[cur ^ 1].data();
		g.rst = rst[cur].data(); g.ast = ast[cur].data(); g.no_stamps = it == 0; // as the device does: the first sweep does not read stamps
		uint32_t acts = 0, tips = 0, ovf = 0, back = 0;
		if (mode == 3) {
			std::vector<cl_binfo_t> tab(cap);
			std::vector<uint32_t> aux(2 * (size_t)cap);
			for (uint32_t k = 0; k < cap; ++k) tab[k].key = CL_NONE;
			cl_bscratch_t b; b.tab = tab.data(); b.used = aux.data(); b.stack = aux.data() + cap; b.cap = cap; b.n_used = 0;
			for (uint32_t v0 = 0; v0 < V; ++v0) {
				if ((uint32_t)G.idx[v0] < 2) continue;
				uint32_t sink = 0, nt = 0;
				int r = cl_bubble_probe(&g, v0, (uint32_t)param, &b, &sink, &nt);
				if (r > 0) { back += cl_bubble_stamp(&g, s, v0, sink, &b); ++acts; tips += nt; }
				else if (r < 0) ovf = 1;
			}
		} else {
			for (uint32_t v = 0; v < V; ++v)
				acts += mode == 0 ? cl_rule_tip(&g, s, v, param) : mode == 1 ? cl_rule_internal(&g, s, v, param) : cl_rule_biloop(&g, s, v, param);
		}
		if (ovf) { cap <<= 2; continue; }
		const bool same = rst[0] == rst[1] && ast[0] == ast[1];
		cur ^= 1;
		*iters = it + 1;
		if (same) {
			*cnt = acts; *cnt2 = tips;
			if (back) return clh_bubble_seq(param, n_seq, n_arc, arc, idx, seq, cnt, cnt2) == 0 ? 1 : -4; /* a pop brings a dead read back (clean_core.h, ASSUMPTION): the sequential sweep, as the device does; 1 = that road was taken */
			break;
		}
	}
	for (uint32_t r = 0; r < n_seq; ++r) if (rst[cur][r] != CL_NONE) seq[r] |= 0x80000000u;
	for (uint32_t e = 0; e < n_arc; ++e) if (ast[cur][e] != CL_NONE) arc[e].ol |= 0x80000000u;
	return 0;
}

// unitigs of a clean graph; outputs sized by the caller: per-unitig arrays [2*n_seq], members [2*n_seq], uarcs [n_arc]
// returns 0 (symmetric graph: chains), 1 (asymmetric: the sequential sweep), -2 (a walk that never ends), -3 (more than mem_cap members)
static int clh_ug_cap(uint32_t n_seq, uint32_t n_arc, const arc_t *arc, const uint64_t *idx, const uint32_t *seq,
                      uint32_t *n_utg, uint32_t *n_mem, uint32_t *n_uarc,
                      uint32_t *u_n, uint32_t *u_len, uint32_t *u_start, uint32_t *u_end, uint32_t *u_off, uint64_t *members, arc_t *uarcs, size_t mem_cap)
{
	Soa G;
	to_soa(n_seq, n_arc, arc, idx, seq, G);
	const uint32_t V = 2 * n_seq;
	*n_utg = *n_mem = *n_uarc = 0;
	if (V == 0) return 0;
	std::vector<uint32_t> nxt(V), prv(V), wt(V), cm(V, UG_NONE), tail(V, UG_NONE), uid(V), flag(V, 0), pos(V + 1, 0), ptr[2], mn[2], dist[2], ws[2];
	std::vector<uint32_t> uh(V), un(V), ul(V, 0), us(V), ue(V), uo(V + 1, 0);
	std::vector<uint8_t> circ(V, 0), ishead(V, 0);
	std::vector<int32_t> mark(V, -1);
	std::vector<unsigned long long> ua(V + 1);
	for (int k = 0; k < 2; ++k) ptr[k].resize(V), mn[k].resize(V), dist[k].resize(V), ws[k].resize(V);
	ug_t a;
	a.au = G.au.data(); a.av = G.av.data(); a.alen = G.alen.data(); a.aol = G.aol.data(); a.idx = G.idx.data(); a.sdel = G.sdel.data(); a.slen = G.slen.data(); a.n_vtx = V;
	a.nxt = nxt.data(); a.prv = prv.data(); a.wt = wt.data(); a.cm = cm.data(); a.tail = tail.data(); a.uid = uid.data(); a.flag = flag.data(); a.pos = pos.data(); a.circ = circ.data(); a.mark = mark.data();
	a.u_head = uh.data(); a.u_n = un.data(); a.u_len = ul.data(); a.u_start = us.data(); a.u_end = ue.data(); a.u_off = uo.data(); a.ua = ua.data();
	int bits = 0; for (uint32_t x = V; x; x >>= 1) ++bits;
	auto rb = [&](int g) { ug_rank_t r; r.ptr = ptr[g].data(); r.mn = mn[g].data(); r.dist = dist[g].data(); r.ws = ws[g].data(); return r; };
	auto rank = [&]() {
		int g = 0;
		for (uint32_t w = 0; w < V; ++w) ugk_jump_init(&a, w, rb(0));
		for (int k = bits + 1; k > 0; --k, g ^= 1)
			for (uint32_t w = 0; w < V; ++w) ugk_jump(w, rb(g), rb(g ^ 1));
		return g;
	};
	for (uint32_t w = 0; w < V; ++w) ugk_link(&a, w);
	int g = rank();
	bool cyc = false, bad = false;
	for (uint32_t w = 0; w < V; ++w) cyc = cyc || (prv[w] < UG_OUT && prv[ptr[g][w]] != UG_NONE), bad = bad || ugk_link_bad(&a, w);
	if (bad) { // not a symmetric graph: the reference's sweep (k_ug_seq on the device); members may overlap, so the caller's arrays bound them
		ug_seq_t s;
		std::vector<uint8_t> seen(V, 0);
		ua.resize(mem_cap + 1);
		a.ua = ua.data();
		s.seen = seen.data(); s.cap = mem_cap; s.n_mem = 0; s.n_utg = 0; s.err = 0;
		for (uint32_t v = 0; v < V && !s.err; ++v)
			if (!G.sdel[v >> 1] && (uint32_t)G.idx[v] > 0 && !seen[v]) ug_seq_unitig(&a, &s, v);
		if (s.err) return -2;
		if (s.n_mem > mem_cap) return -3;
		const uint32_t U = s.n_utg, M = (uint32_t)s.n_mem;
		*n_utg = U; *n_mem = M;
		for (uint32_t k = 0; k < U; ++k) ugk_mark(&a, k);
		uint32_t na = 0;
		for (uint32_t e = 0; e < n_arc; ++e)
			if (ugk_arc_keep(&a, e)) { uint32_t o[4]; ugk_arc_emit(&a, e, o); uarcs[na].len = o[0]; uarcs[na].u = o[1]; uarcs[na].v = o[2]; uarcs[na].ol = o[3]; ++na; }
		*n_uarc = na;
		memcpy(u_n, un.data(), U * 4); memcpy(u_len, ul.data(), U * 4); memcpy(u_start, us.data(), U * 4); memcpy(u_end, ue.data(), U * 4); memcpy(u_off, uo.data(), U * 4);
		memcpy(members, ua.data(), (size_t)M * 8);
		return 1;
	}
	if (cyc) { // as the device: cut + second ranking only when some chain has no head
		for (uint32_t w = 0; w < V; ++w) ishead[w] = prv[w] == UG_NONE;
		for (uint32_t w = 0; w < V; ++w) ugk_cut(&a, w, ptr[g].data(), mn[g].data(), ishead.data());
		g = rank();
	}
	const ug_rank_t RK = rb(g);
	const uint32_t *P = RK.ptr, *D = RK.dist;
	for (uint32_t w = 0; w < V; ++w) ugk_chain(&a, w, RK);
	for (uint32_t w = 0; w < V; ++w) ugk_pick(&a, w, P);
	uint32_t U = 0;
	for (uint32_t w = 0; w < V; ++w) { pos[w] = U; U += flag[w]; }
	*n_utg = U;
	if (U == 0) return 0;
	for (uint32_t w = 0; w < V; ++w) ugk_units(&a, w, RK);
	uint32_t M = 0;
	for (uint32_t k = 0; k < U; ++k) { uo[k] = M; M += un[k]; }
	*n_mem = M;
	for (uint32_t w = 0; w < V; ++w) ugk_fill(&a, w, P, D);
	for (uint32_t k = 0; k < U; ++k) ugk_mark(&a, k);
	uint32_t na = 0;
	for (uint32_t e = 0; e < n_arc; ++e)
		if (ugk_arc_keep(&a, e)) { uint32_t o[4]; ugk_arc_emit(&a, e, o); uarcs[na].len = o[0]; uarcs[na].u = o[1]; uarcs[na].v = o[2]; uarcs[na].ol = o[3]; ++na; }
	*n_uarc = na;
	memcpy(u_n, un.data(), U * 4); memcpy(u_len, ul.data(), U * 4); memcpy(u_start, us.data(), U * 4); memcpy(u_end, ue.data(), U * 4); memcpy(u_off, uo.data(), U * 4);
	memcpy(members, ua.data(), (size_t)M * 8);
	return 0;
}

extern "C" int clh_ug(uint32_t n_seq, uint32_t n_arc, const arc_t *arc, const uint64_t *idx, const uint32_t *seq,
                      uint32_t *n_utg, uint32_t *n_mem, uint32_t *n_uarc,
                      uint32_t *u_n, uint32_t *u_len, uint32_t *u_start, uint32_t *u_end, uint32_t *u_off, uint64_t *members, arc_t *uarcs)
{
	return clh_ug_cap(n_seq, n_arc, arc, idx, seq, n_utg, n_mem, n_uarc, u_n, u_len, u_start, u_end, u_off, members, uarcs, 2 * (size_t)n_seq);
}
// the same with room for mem_cap members (asymmetric graphs: unitigs may share reads)
extern "C" int clh_ug2(uint32_t n_seq, uint32_t n_arc, const arc_t *arc, const uint64_t *idx, const uint32_t *seq,
                       uint32_t *n_utg, uint32_t *n_mem, uint32_t *n_uarc,
                       uint32_t *u_n, uint32_t *u_len, uint32_t *u_start, uint32_t *u_end, uint32_t *u_off, uint64_t *members, arc_t *uarcs, size_t mem_cap)
{
	return clh_ug_cap(n_seq, n_arc, arc, idx, seq, n_utg, n_mem, n_uarc, u_n, u_len, u_start, u_end, u_off, members, uarcs, mem_cap);
}

// mcs_capi_match.hip — matcher half of the C ABI: top-K Hamming lists and the three brute-force searches of cORBmatcher.
#include "mcs_host.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace mcs {
void launch_single_distance(const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out, hipStream_t s);
void launch_rows_valid(const int* nkp, int nimg, int cap, uint8_t* valid, hipStream_t s);
void launch_rig_pack_headers(const int* nkp, int nimg, int cap, uint8_t* blocks, int rowStride, hipStream_t s);
void launch_rig_rows_valid(const uint8_t* blocks, int nimg, int cap, int rowStride, uint8_t* valid, int* nkpOut, hipStream_t s);
}
using namespace mcs;

static int ensure(void** p, size_t* cap, size_t need) {
	if (*cap >= need && *p) return MCS_OK;
	if (*p) (void)hipFree(*p);
	*p = nullptr; *cap = 0;
	size_t want = need + need / 2 + 256;
	HIPCHK(hipMalloc(p, want));
	*cap = want;
	return MCS_OK;
}

struct DevSets {   // device views of a (query sets, train sets) pair
	const uint8_t *qd, *qm, *qvalid; const int* qgroup;
	const uint8_t *td, *tm, *tvalid; const int* tgroup;
};

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

// How the nsets (query set, train set) pairs of one call map onto the caller's arrays: pair s reads query set (s % qmod) and train set (s / tdiv);
// nq_sets / nt_sets = number of distinct sets behind each pointer (what a host-kind call has to stage).
struct SetGrid { int nsets, qmod, tdiv, nq_sets, nt_sets; int toff = 0, tmod = 0x7FFFFFFF; };
static size_t set_span(const mcs_desc_set* d) {   // rows from a set's first to one past its last row
	if (d->block_rows == 0 || d->n == 0) return (size_t)d->n;
	return (size_t)(d->n / d->block_rows - 1) * (size_t)d->block_pitch_rows + (size_t)d->block_rows;
}
static SetGrid grid_batched(int nsets, size_t qpitch, size_t tpitch) { SetGrid g; g.nsets = nsets; g.qmod = nsets; g.tdiv = 1; g.nq_sets = qpitch ? nsets : 1; g.nt_sets = tpitch ? nsets : 1; return g; }
static SetGrid grid_sweep(int nq_sets, int nt_sets) { SetGrid g; g.nsets = nq_sets * nt_sets; g.qmod = nq_sets; g.tdiv = nq_sets; g.nq_sets = nq_sets; g.nt_sets = nt_sets; return g; }
// pairs (frame first + s, its predecessor in a ring of `total` frames), s = 0 .. count-1; both pointers at frame 0 of the ring
static SetGrid grid_ring(int total, int first, int count) { SetGrid g; g.nsets = count; g.qmod = count; g.tdiv = 1; g.nq_sets = total; g.nt_sets = total; g.toff = first + total - 1; g.tmod = total; return g; }

static int validate_sets(mcs_ctx* c, int nsets, const mcs_desc_set* q, const mcs_desc_set* t, int dim, int K) {
	if (!c || !q || !t) return fail(MCS_ERR_INVALID, "null argument");
	if (dim != 16 && dim != 32 && dim != 64) return fail(MCS_ERR_INVALID, "dim must be 16, 32 or 64");
	if (K != 1 && K != 2 && K != 4 && K != 8 && K != 16 && K != 32) return fail(MCS_ERR_INVALID, "K must be 1,2,4,8,16 or 32");
	if (nsets < 1 || q->n < 0 || t->n < 0 || t->n >= (1 << 20)) return fail(MCS_ERR_INVALID, "bad set size (train rows must be < 2^20)");
	if (q->stride < dim || t->stride < dim || (q->stride & 3) || (t->stride & 3)) return fail(MCS_ERR_INVALID, "descriptor stride must be >= dim and a multiple of 4");
	if ((q->mask == nullptr) != (t->mask == nullptr)) return fail(MCS_ERR_INVALID, "masks must be given for both sets or neither");
	if ((q->n > 0 && !q->desc) || (t->n > 0 && !t->desc)) return fail(MCS_ERR_INVALID, "null descriptors");
	for (const mcs_desc_set* d : {q, t})
		if (d->block_rows != 0 && (d->block_rows < 1 || d->n % d->block_rows != 0 || d->block_pitch_rows < d->block_rows))
			return fail(MCS_ERR_INVALID, "block_rows must divide n and block_pitch_rows must be >= block_rows");
	return MCS_OK;
}

namespace mcs {
__global__ __launch_bounds__(256) void k_search_out(int* __restrict__ dm, const int* __restrict__ sm, size_t n, int* __restrict__ dn, const int* __restrict__ sn, int* __restrict__ df,
                                                    const int* __restrict__ sf, int nsets) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dm[i] = sm[i];
	if (blockIdx.x == 0)
		for (int i = threadIdx.x; i < nsets; i += 256) { dn[i] = sn[i]; if (df) df[i] = sf[i]; }
}
}  // namespace mcs
static void launch_search_out(int* dm, const int* sm, size_t n, int* dn, const int* sn, int* df, const int* sf, int nsets, hipStream_t s) {
	const int blocks = (int)std::min<size_t>(std::max<size_t>((n + 1023) / 1024, 1), 64);
	hipLaunchKernelGGL(mcs::k_search_out, dim3(blocks), dim3(256), 0, s, dm, sm, n, dn, sn, df, sf, nsets);
}

// host pointers -> staged device copies (on the context's stream); device pointers pass through
static int stage_sets(mcs_ctx* c, const SetGrid& sg, const mcs_desc_set* q, size_t qpitch, const mcs_desc_set* t, size_t tpitch, mcs_mem_kind kind,
                      DevSets* out, const double** rays1, const double** rays2, const double** E, size_t nE) {
	if (kind == MCS_MEM_DEVICE) {
		out->qd = q->desc; out->qm = q->mask; out->qvalid = q->valid; out->qgroup = q->group;
		out->td = t->desc; out->tm = t->mask; out->tvalid = t->valid; out->tgroup = t->group;
		return MCS_OK;
	}
	hipStream_t s = c->stream;
	const size_t qRows = qpitch * (sg.nq_sets - 1) + set_span(q), tRows = tpitch * (sg.nt_sets - 1) + set_span(t);
	size_t need = 0;
	const size_t oQd = need; need += al256(qRows * q->stride);
	// descriptor | mask interleaved in one row (the rig's exchange blocks): the mask rides along with the descriptor copy
	const bool qInter = q->mask && q->mask > q->desc && q->mask - q->desc < q->stride, tInter = t->mask && t->mask > t->desc && t->mask - t->desc < t->stride;
	const size_t oQm = need; need += (q->mask && !qInter) ? al256(qRows * q->stride) : 0;
	const size_t oQv = need; need += q->valid ? al256(qRows) : 0;
	const size_t oQg = need; need += q->group ? al256(qRows * 4) : 0;
	const size_t oTd = need; need += al256(tRows * t->stride);
	const size_t oTm = need; need += (t->mask && !tInter) ? al256(tRows * t->stride) : 0;
	const size_t oTv = need; need += t->valid ? al256(tRows) : 0;
	const size_t oTg = need; need += t->group ? al256(tRows * 4) : 0;
	const size_t oR1 = need; need += (rays1 && *rays1) ? al256(qRows * 24) : 0;
	const size_t oR2 = need; need += (rays2 && *rays2) ? al256(tRows * 24) : 0;
	const size_t oE = need; need += (E && *E) ? al256(nE * 8) : 0;
	HIPCHK(hipStreamSynchronize(s));   // staging buffer may still be in use by an earlier call
	if (int r = ensure((void**)&c->stage, &c->stageCap, need)) return r;
	uint8_t* st = c->stage;
	PinnedUpload up;   // all inputs in one H2D copy (every host-kind search ends with a stream synchronisation)
	HIPCHK(up.begin(c, st, need));
	if (qRows) up.put(oQd, q->desc, qRows * q->stride);
	if (q->mask && !qInter && qRows) up.put(oQm, q->mask, qRows * q->stride);
	if (q->valid && qRows) up.put(oQv, q->valid, qRows);
	if (q->group && qRows) up.put(oQg, q->group, qRows * 4);
	if (tRows) up.put(oTd, t->desc, tRows * t->stride);
	if (t->mask && !tInter && tRows) up.put(oTm, t->mask, tRows * t->stride);
	if (t->valid && tRows) up.put(oTv, t->valid, tRows);
	if (t->group && tRows) up.put(oTg, t->group, tRows * 4);
	out->qd = st + oQd; out->qm = q->mask ? (qInter ? st + oQd + (q->mask - q->desc) : st + oQm) : nullptr; out->qvalid = q->valid ? st + oQv : nullptr;
	out->qgroup = q->group ? (const int*)(st + oQg) : nullptr;
	out->td = st + oTd; out->tm = t->mask ? (tInter ? st + oTd + (t->mask - t->desc) : st + oTm) : nullptr; out->tvalid = t->valid ? st + oTv : nullptr;
	out->tgroup = t->group ? (const int*)(st + oTg) : nullptr;
	if (rays1 && *rays1) { if (qRows) up.put(oR1, *rays1, qRows * 24); *rays1 = (const double*)(st + oR1); }
	if (rays2 && *rays2) { if (tRows) up.put(oR2, *rays2, tRows * 24); *rays2 = (const double*)(st + oR2); }
	if (E && *E) { up.put(oE, *E, nE * 8); *E = (const double*)(st + oE); }
	HIPCHK(up.flush(s));
	return MCS_OK;
}

static int run_topk(mcs_ctx* c, const DevSets& d, const SetGrid& sg, const mcs_desc_set* q, size_t qpitch, const mcs_desc_set* t, size_t tpitch, int dim,
                    int K, int count_thresh, int max_dist, int* outDist, int* outIdx, int* outCount, hipStream_t ls = nullptr, int slot = 0) {
	MatchArgs a{};
	const int nsets = sg.nsets;
	a.maxDist = max_dist;
	// deferred (ls = the greedy stream): the lists run there, ordered behind the previous search by the stream itself and behind the caller's stream by
	// an event recorded AFTER the train sets' pass, which stays on the caller's stream — 30 us there, and the matcher starts together with whatever the
	// caller enqueues next (started 40 us later it found the chip full of the next batch's FAST workgroups: 0.72 instead of 0.30 ms)
	const bool deferred = ls != nullptr;
	if (!deferred) ls = c->stream;
	if (!deferred && c->side && c->greedyPending) { HIPCHK(hipStreamWaitEvent(c->stream, c->evGreedy, 0)); c->greedyPending = false; }
	if (int r = ensure((void**)(slot ? &c->topKeys2 : &c->topKeys), slot ? &c->topKeys2Cap : &c->topKeysCap, std::max<size_t>((size_t)nsets * q->n, 1) * K * sizeof(uint32_t))) return r;
	a.keys = slot ? c->topKeys2 : c->topKeys;
	a.qd = d.qd; a.qm = d.qm; a.qvalid = d.qvalid; a.qgroup = d.qgroup; a.td = d.td; a.tm = d.tm; a.tvalid = d.tvalid; a.tgroup = d.tgroup;
	a.nq = q->n; a.nt = t->n; a.qstride = q->stride; a.tstride = t->stride; a.qpitch = qpitch; a.tpitch = tpitch;
	a.nsets = nsets; a.qmod = sg.qmod; a.tdiv = sg.tdiv; a.dim = dim; a.K = K; a.countThresh = count_thresh;
	a.qblk = q->block_rows; a.qbpitch = (size_t)q->block_pitch_rows; a.tblk = t->block_rows; a.tbpitch = (size_t)t->block_pitch_rows;
	a.toff = sg.toff; a.tmod = sg.tmod;
	const size_t outRows = (size_t)nsets * q->n;
	const int qTiles = (q->n + 255) / 256;
	// Train-range splits only where the (set, query tile) grid alone cannot fill the chip (a single keyframe pair: 12 workgroups).  From kFillBlocks
	// workgroups on (2 per CU) every workgroup walks its whole train set: no partial lists, no merge pass (their HBM traffic was 16x the algorithmic bytes).
	static const int kTargetBlocks = getenv("MCS_MATCH_BLOCKS") ? atoi(getenv("MCS_MATCH_BLOCKS")) : 2048;
	static const int kFillBlocks = getenv("MCS_MATCH_FILL") ? atoi(getenv("MCS_MATCH_FILL")) : 512;
	int splits = (long long)qTiles * nsets >= kFillBlocks ? 1 : (int)((kTargetBlocks + (long long)qTiles * nsets - 1) / ((long long)qTiles * nsets));
	splits = std::max(1, std::min(splits, (t->n + 255) / 256));
	a.splits = splits;
	if (splits > 1) {
		if (int r = ensure((void**)&c->partial, &c->partialCap, outRows * splits * K * sizeof(uint32_t))) return r;
		if (int r = ensure((void**)&c->partialCount, &c->partialCountCap, outRows * splits * sizeof(int))) return r;
	}
	a.partial = c->partial; a.partialCount = c->partialCount;
	a.outDist = outDist; a.outIdx = outIdx; a.outCount = outCount;
	if (match_mfma_shape(a) && t->n > 0) {
		// the matrix-core matcher reads train sets that one pass has compacted and expanded (256 bytes per masked 32-byte row); beyond kExpandCap the v_bcnt kernel serves
		static const size_t kExpandCap = getenv("MCS_MATCH_EXPAND_MB") ? (size_t)atoll(getenv("MCS_MATCH_EXPAND_MB")) << 20 : (size_t)8 << 30;
		size_t bA = 0, bW = 0;
		match_mfma_scratch(a, sg.nt_sets, &bA, &bW, &a.exStages);
		if (bA <= kExpandCap) {
			if (int r = ensure((void**)&c->exA, &c->exACap, bA)) return r;
			if (int r = ensure((void**)&c->exW, &c->exWCap, bW)) return r;
			if (int r = ensure((void**)&c->exRows, &c->exRowsCap, (size_t)sg.nt_sets * sizeof(int))) return r;
			a.exA = (uint4*)c->exA; a.exW = (float*)c->exW; a.exRows = c->exRows; a.tsets = sg.nt_sets;
		}
	}
	c->tic("match");
	static const bool valuOnly = getenv("MCS_MATCH_VALU") != nullptr;
	if (deferred) {
		if (!valuOnly && match_mfma_serves(a)) {
			// the previous deferred search's LISTS (other stream) may still read the expanded sets: order the pass behind them (in a pipelined caller they finished long ago)
			if (c->searchSeq > 0) HIPCHK(hipStreamWaitEvent(c->stream, c->evLists, 0));
			launch_match_expand(a, c->stream); a.exDone = 1;
		}
		HIPCHK(hipEventRecord(c->evMatch, c->stream));
		HIPCHK(hipStreamWaitEvent(ls, c->evMatch, 0));
		if (c->searchSeq > 1) HIPCHK(hipStreamWaitEvent(ls, c->evGreedyBuf[slot], 0));   // the greedy pass that read this list buffer (two searches ago)
	}
	launch_match(a, ls);
	if (deferred) HIPCHK(hipEventRecord(c->evLists, ls));
	c->toc("match");
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

extern "C" {

int mcs_match_topk_batched(mcs_ctx* c, int nsets, const mcs_desc_set* q, size_t qpitch, const mcs_desc_set* t, size_t tpitch, int dim, int K,
                           int count_thresh, mcs_mem_kind kind, int32_t* out_dist, int32_t* out_idx, int32_t* out_count_le) {
	if (int r = validate_sets(c, nsets, q, t, dim, K)) return r;
	if (!out_dist || !out_idx || !out_count_le) return fail(MCS_ERR_INVALID, "null output");
	if (q->n == 0) return MCS_OK;
	HIPCHK(hipSetDevice(c->device));
	DevSets d{};
	const SetGrid sg = grid_batched(nsets, qpitch, tpitch);
	if (int r = stage_sets(c, sg, q, qpitch, t, tpitch, kind, &d, nullptr, nullptr, nullptr, 0)) return r;
	const size_t outRows = (size_t)nsets * q->n;
	if (kind == MCS_MEM_DEVICE) return run_topk(c, d, sg, q, qpitch, t, tpitch, dim, K, count_thresh, 0x7FFFFFFF, out_dist, out_idx, out_count_le);
	const size_t oD = 0, oI = al256(outRows * K * 4), oC = oI + al256(outRows * K * 4);
	if (int r = ensure((void**)&c->stageOut, &c->stageOutCap, oC + al256(outRows * 4))) return r;
	uint8_t* so = c->stageOut;
	if (int r = run_topk(c, d, sg, q, qpitch, t, tpitch, dim, K, count_thresh, 0x7FFFFFFF, (int*)(so + oD), (int*)(so + oI), (int*)(so + oC))) return r;
	HIPCHK(hipMemcpyAsync(out_dist, so + oD, outRows * K * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out_idx, so + oI, outRows * K * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out_count_le, so + oC, outRows * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return MCS_OK;
}

int mcs_match_topk(mcs_ctx* c, const mcs_desc_set* q, const mcs_desc_set* t, int dim, int K, int count_thresh, mcs_mem_kind kind,
                   int32_t* out_dist, int32_t* out_idx, int32_t* out_count_le) {
	return mcs_match_topk_batched(c, 1, q, 0, t, 0, dim, K, count_thresh, kind, out_dist, out_idx, out_count_le);
}

// mode 0: SearchByBoW(KF,KF)   1: SearchByBoW(KF,F)   2: SearchForTriangulationRaw
static int search_common(mcs_ctx* c, int mode, const SetGrid& sg, const mcs_desc_set* q, size_t qpitch, const mcs_desc_set* t, size_t tpitch, int dim,
                         double nnratio, int K, mcs_mem_kind kind, const double* rays1, const double* rays2, const double* E, size_t Epitch, int nrCams,
                         int32_t* out_match, int32_t* out_nmatches, int32_t* out_fallbacks) {
	const int nsets = sg.nsets;
	if (sg.nq_sets < 1 || sg.nt_sets < 1) return fail(MCS_ERR_INVALID, "bad set count");
	if (int r = validate_sets(c, nsets, q, t, dim, K)) return r;
	if (!out_match || !out_nmatches) return fail(MCS_ERR_INVALID, "null output");
	if (t->n > 131072) return fail(MCS_ERR_UNSUPPORTED, "more than 131072 train rows per set");
	if (mode == 2 && (!rays1 || !rays2 || !E || nrCams < 1)) return fail(MCS_ERR_INVALID, "triangulation search needs rays and essential matrices");
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	// Deferred form (mcs_ctx_set_async_search, device memory): lists AND greedy pass run on the greedy stream, ordered behind the inputs by an event and
	// behind the previous search by the stream itself; the caller's stream goes on at once (the next batch's extraction fills the matcher's stalls).
	const bool deferred = kind == MCS_MEM_DEVICE && c->overlap() && c->asyncSearch;
	if (deferred) {
		if (q->n == 0) { HIPCHK(hipEventRecord(c->evMatch, s)); HIPCHK(hipStreamWaitEvent(c->side2, c->evMatch, 0)); }   // otherwise in run_topk
	} else if (c->side && c->greedyPending) {   // the previous search's greedy pass (side stream) still reads the shared list buffers
		HIPCHK(hipStreamWaitEvent(s, c->evGreedy, 0));
		c->greedyPending = false;
	}
	const bool havingMasks = q->mask != nullptr;
	// thresholds of cORBmatcher::cORBmatcher (src/cORBmatcher.cpp:46-65); only TH_LOW_ is used by these three searches
	const int thLow = havingMasks ? (int)floor((double)dim) : 2 * dim;
	DevSets d{};
	if (int r = stage_sets(c, sg, q, qpitch, t, tpitch, kind, &d, mode == 2 ? &rays1 : nullptr, mode == 2 ? &rays2 : nullptr,
	                       mode == 2 ? &E : nullptr, Epitch * (size_t)(nsets - 1) + (size_t)nrCams * nrCams * 9)) return r;
	const size_t rows = (size_t)nsets * q->n;
	// deferred: two list buffers in turn, the greedy pass on a stream of its own — it is a chain of dependent round trips with a few waves per CU, the matcher waits
	// 45 % of its cycles: side by side (greedy pass of search n, lists of search n + 1) the greedy pass disappears from the matcher stream's critical path
	const int slot = deferred ? (int)(c->searchSeq & 1) : 0;
	if (int r = ensure((void**)&c->topCnt, &c->topCntCap, std::max<size_t>(rows, 1) * 4)) return r;
	{
		// Rows that can never influence a decision stay out of the lists: SearchByBoW needs best <= TH_LOW and, for the ratio
		// test, seconds up to the largest d with nnratio*d <= TH_LOW (any farther second passes best < nnratio*second for every
		// admissible best); SearchForTriangulationRaw only collects dist <= TH_LOW (:1062).
		int maxDist = thLow;
		if (mode != 2) while (maxDist < 8 * dim && nnratio * static_cast<double>(maxDist + 1) <= static_cast<double>(thLow)) ++maxDist;
		if (q->n > 0)
			if (int r = run_topk(c, d, sg, q, qpitch, t, tpitch, dim, K, -1, maxDist, nullptr, nullptr, c->topCnt, deferred ? c->side2 : nullptr, slot)) return r;
	}
	GreedyArgs g{};
	g.qd = d.qd; g.qm = d.qm; g.qvalid = d.qvalid; g.qgroup = d.qgroup; g.td = d.td; g.tm = d.tm; g.tvalid = d.tvalid; g.tgroup = d.tgroup;
	g.nq = q->n; g.nt = t->n; g.qstride = q->stride; g.tstride = t->stride; g.qpitch = qpitch; g.tpitch = tpitch;
	g.nsets = nsets; g.qmod = sg.qmod; g.tdiv = sg.tdiv; g.dim = dim; g.K = K; g.keys = slot ? c->topKeys2 : c->topKeys;
	g.qblk = q->block_rows; g.qbpitch = (size_t)q->block_pitch_rows; g.tblk = t->block_rows; g.tbpitch = (size_t)t->block_pitch_rows;
	g.toff = sg.toff; g.tmod = sg.tmod;
	g.thLow = thLow; g.thInclusive = mode == 1 ? 1 : 0; g.ratio = nnratio; g.mode = mode;
	g.rays1 = rays1; g.rays2 = rays2; g.E = E; g.Epitch = Epitch; g.nrCams = nrCams;
	{ static const int ms = getenv("MCS_JACOBI_MAX_SWEEPS") ? atoi(getenv("MCS_JACOBI_MAX_SWEEPS")) : 0; g.jacMaxSweeps = ms; }
	const size_t outN = (size_t)nsets * (mode == 1 ? t->n : q->n);
	if (kind == MCS_MEM_DEVICE) {
		g.outMatch = out_match; g.outCount = out_nmatches; g.outFallbacks = out_fallbacks;
		if (deferred) {
			if (q->n > 0) HIPCHK(hipStreamWaitEvent(c->side3, c->evLists, 0));
			else HIPCHK(hipStreamWaitEvent(c->side3, c->evMatch, 0));
			launch_greedy(g, c->side3);
			HIPCHK(hipEventRecord(c->evGreedyBuf[slot], c->side3));
			HIPCHK(hipEventRecord(c->evGreedy, c->side3));
			HIPCHK(hipEventRecord(c->evSearch[c->searchSeq & 3], c->side3));
			++c->searchSeq;
			c->greedyPending = true;
			c->lastResultStream = c->side3;
		} else if (c->overlap()) {
			// the greedy resolution is one wave per set pair (latency-bound): run it on the side stream so that whatever the caller
			// enqueues next on the main stream (the next batch's extraction) fills the idle CUs.  mcs_ctx_join / the next search /
			// mcs_ctx_synchronize order later work behind it.
			HIPCHK(hipEventRecord(c->evMatch, s));
			HIPCHK(hipStreamWaitEvent(c->side2, c->evMatch, 0));
			launch_greedy(g, c->side2);
			HIPCHK(hipEventRecord(c->evGreedy, c->side2));
			c->greedyPending = true;
			c->lastResultStream = c->side2;
		} else { c->tic("greedy"); launch_greedy(g, s); c->toc("greedy"); c->lastResultStream = s; }
		HIPCHK(hipGetLastError());
		return MCS_OK;
	}
	const size_t oM = 0, oN = al256(std::max<size_t>(outN, 1) * 4), oF = oN + al256((size_t)nsets * 4);
	if (int r = ensure((void**)&c->stageOut, &c->stageOutCap, oF + al256((size_t)nsets * 4))) return r;
	uint8_t* so = c->stageOut;
	g.outMatch = (int*)(so + oM); g.outCount = (int*)(so + oN); g.outFallbacks = (int*)(so + oF);
	c->tic("greedy"); launch_greedy(g, s); c->toc("greedy");
	HIPCHK(hipGetLastError());
	{
		// page-locked outputs: one launch instead of three copies (mcs_host.h device_view)
		static const bool outKernel = !(getenv("MCS_OUT_KERNEL") && atoi(getenv("MCS_OUT_KERNEL")) == 0);
		int* dm = (int*)device_view(out_match); int* dn = (int*)device_view(out_nmatches); int* df = (int*)device_view(out_fallbacks);
		if (outKernel && dm && dn && (df || !out_fallbacks)) {
			launch_search_out(dm, (const int*)(so + oM), outN, dn, (const int*)(so + oN), df, (const int*)(so + oF), nsets, s);
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(s));
			return MCS_OK;
		}
	}
	if (outN) HIPCHK(hipMemcpyAsync(out_match, so + oM, outN * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(out_nmatches, so + oN, (size_t)nsets * 4, hipMemcpyDeviceToHost, s));
	if (out_fallbacks) HIPCHK(hipMemcpyAsync(out_fallbacks, so + oF, (size_t)nsets * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	return MCS_OK;
}

int mcs_search_kf_kf(mcs_ctx* c, int nsets, const mcs_desc_set* kf1, size_t pitch1, const mcs_desc_set* kf2, size_t pitch2, int dim, double nnratio,
                     int K, mcs_mem_kind kind, int32_t* match12, int32_t* nmatches, int32_t* fallbacks) {
	return search_common(c, 0, grid_batched(nsets, pitch1, pitch2), kf1, pitch1, kf2, pitch2, dim, nnratio, K, kind, nullptr, nullptr, nullptr, 0, 0, match12, nmatches, fallbacks);
}

int mcs_search_kf_kf_ring(mcs_ctx* c, int nframes_total, int first, int count, const mcs_desc_set* frames, size_t pitch_rows, int dim, double nnratio, int K,
                          mcs_mem_kind kind, int32_t* match12, int32_t* nmatches, int32_t* fallbacks) {
	if (nframes_total < 2 || first < 0 || count < 1 || first + count > nframes_total) return fail(MCS_ERR_INVALID, "bad ring range");
	if (kind != MCS_MEM_DEVICE) return fail(MCS_ERR_UNSUPPORTED, "the ring form takes device memory");
	if (!frames) return fail(MCS_ERR_INVALID, "null argument");
	// the query side starts at frame `first`: shift its pointers, the train side stays at frame 0 and is addressed through (first + s - 1) mod total
	mcs_desc_set q = *frames;
	const size_t shift = (size_t)first * pitch_rows;
	q.desc += shift * frames->stride;
	if (q.mask) q.mask += shift * frames->stride;
	if (q.valid) q.valid += shift;
	if (q.group) q.group += shift;
	return search_common(c, 0, grid_ring(nframes_total, first, count), &q, pitch_rows, frames, pitch_rows, dim, nnratio, K, kind, nullptr, nullptr, nullptr, 0, 0, match12, nmatches, fallbacks);
}

int mcs_search_kf_f(mcs_ctx* c, int nsets, const mcs_desc_set* kf, size_t pitchKF, const mcs_desc_set* f, size_t pitchF, int dim, double nnratio, int K,
                    mcs_mem_kind kind, int32_t* matchF, int32_t* nmatches, int32_t* fallbacks) {
	return search_common(c, 1, grid_batched(nsets, pitchKF, pitchF), kf, pitchKF, f, pitchF, dim, nnratio, K, kind, nullptr, nullptr, nullptr, 0, 0, matchF, nmatches, fallbacks);
}

int mcs_search_kf_f_sweep(mcs_ctx* c, int nkf, const mcs_desc_set* kf, size_t pitchKF, int nframes, const mcs_desc_set* f, size_t pitchF, int dim,
                          double nnratio, int K, mcs_mem_kind kind, int32_t* matchF, int32_t* nmatches, int32_t* fallbacks) {
	if (nkf < 1 || nframes < 1 || (long long)nkf * nframes > (1 << 24)) return fail(MCS_ERR_INVALID, "bad keyframe / frame count");
	auto rows_of = [](const mcs_desc_set* d) { return (size_t)(d ? (d->block_rows ? d->block_rows : d->n) : 0); };   // interleaved blocks: only a block must fit
	if ((nkf > 1 && pitchKF < rows_of(kf)) || (nframes > 1 && pitchF < rows_of(f))) return fail(MCS_ERR_INVALID, "set pitch smaller than the set");
	return search_common(c, 1, grid_sweep(nkf, nframes), kf, pitchKF, f, pitchF, dim, nnratio, K, kind, nullptr, nullptr, nullptr, 0, 0, matchF, nmatches, fallbacks);
}

int mcs_search_triangulation(mcs_ctx* c, int nsets, const mcs_desc_set* kf1, size_t pitch1, const mcs_desc_set* kf2, size_t pitch2, const double* rays1,
                             const double* rays2, const double* E, int nrCams, int dim, int K, mcs_mem_kind kind, int32_t* match12,
                             int32_t* nmatches, int32_t* fallbacks) {
	return search_common(c, 2, grid_batched(nsets, pitch1, pitch2), kf1, pitch1, kf2, pitch2, dim, 0.0, K, kind, rays1, rays2, E, 0, nrCams, match12, nmatches, fallbacks);
}

int mcs_search_triangulation_sweep(mcs_ctx* c, int nsets, const mcs_desc_set* kf1, size_t pitch1, const mcs_desc_set* kf2, size_t pitch2, const double* rays1,
                                   const double* rays2, const double* E, size_t E_set_pitch, int nrCams, int dim, int K, mcs_mem_kind kind,
                                   int32_t* match12, int32_t* nmatches, int32_t* fallbacks) {
	if (E_set_pitch != 0 && E_set_pitch < (size_t)nrCams * nrCams * 9) return fail(MCS_ERR_INVALID, "E_set_pitch smaller than one block of essential matrices");
	return search_common(c, 2, grid_batched(nsets, pitch1, pitch2), kf1, pitch1, kf2, pitch2, dim, 0.0, K, kind, rays1, rays2, E, E_set_pitch, nrCams, match12, nmatches, fallbacks);
}

int mcs_search_by_projection(mcs_ctx* c, const mcs_projection_set* mp, const mcs_frame_view* f, double th, double nnratio, int dim, mcs_mem_kind kind,
                             int32_t* match, int32_t* nmatches) {
	if (!c || !mp || !f || !match || !nmatches) return fail(MCS_ERR_INVALID, "null argument");
	if (dim != 16 && dim != 32 && dim != 64) return fail(MCS_ERR_INVALID, "dim must be 16, 32 or 64");
	if (mp->n < 0 || f->n < 0 || f->n > 65536 || f->nr_cams < 1 || f->nlevels < 1) return fail(MCS_ERR_INVALID, "bad sizes (frame features must be <= 65536)");
	if ((mp->mask == nullptr) != (f->mask == nullptr)) return fail(MCS_ERR_INVALID, "masks must be given for both sides or neither");
	if (mp->stride < dim || f->stride < dim || (mp->stride & 3) || (f->stride & 3)) return fail(MCS_ERR_INVALID, "descriptor stride must be >= dim and a multiple of 4");
	if (kind == MCS_MEM_HOST)   // level[] indexes scale_factors on the device
		for (int i = 0; i < mp->n; ++i)
			if (mp->level[i] < 0 || mp->level[i] >= f->nlevels) return fail(MCS_ERR_INVALID, "projection level outside [0, nlevels)");
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	if (c->side && c->greedyPending) { HIPCHK(hipStreamWaitEvent(s, c->evGreedy, 0)); c->greedyPending = false; }
	const bool havingMasks = mp->mask != nullptr;
	ProjArgs a{};
	a.nproj = mp->n; a.pstride = mp->stride; a.nfeat = f->n; a.fstride = f->stride; a.nrCams = f->nr_cams;
	a.th = th; a.ratio = nnratio; a.dim = dim; a.rule = 0; a.cap = kProjListK;
	a.thHigh = havingMasks ? (int)floor(1.5 * dim) : 3 * dim;   // TH_HIGH_ (src/cORBmatcher.cpp:46-65)
	const size_t np = std::max(mp->n, 1), nf = std::max(f->n, 1);
	// scratch: lists + counts (+ staged inputs / outputs for host pointers), one allocation per call (this row is not a bench path)
	size_t need = al256(np * kProjListK * 8) + al256(np * 4);
	const size_t oLists = 0, oCounts = al256(np * kProjListK * 8);
	size_t o = need;
	auto reserve = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
	size_t oPx = 0, oPy = 0, oVc = 0, oLv = 0, oPc = 0, oPd = 0, oPm = 0, oKeys = 0, oFd = 0, oFm = 0, oFc = 0, oAs = 0, oW = 0, oH = 0, oSc = 0, oMatch = 0, oNm = 0;
	const bool host = kind == MCS_MEM_HOST;
	if (host) {
		oPx = reserve(np * 8); oPy = reserve(np * 8); oVc = reserve(np * 8); oLv = reserve(np * 4); oPc = reserve(np * 4);
		oPd = reserve(np * mp->stride); oPm = reserve(np * mp->stride); oKeys = reserve(nf * sizeof(mcs_keypoint));
		oFd = reserve(nf * f->stride); oFm = reserve(nf * f->stride); oFc = reserve(nf * 4); oAs = reserve(nf);
		oW = reserve((size_t)f->nr_cams * 4); oH = reserve((size_t)f->nr_cams * 4); oSc = reserve((size_t)f->nlevels * 8);
		oMatch = reserve(np * 4); oNm = reserve(4);
	}
	uint8_t* buf = nullptr;
	HIPCHK(ctx_arena(c, o, &buf));   // persistent per context (a hipMalloc / hipFree pair per call cost more than the kernels)
	auto done = [&](int rc) { (void)hipStreamSynchronize(s); return rc; };
	a.lists = (unsigned long long*)(buf + oLists); a.counts = (int*)(buf + oCounts);
	if (host) {
		PinnedUpload up;   // one H2D copy for all inputs
		HIPCHK(up.begin(c, buf, o));
#define UP(off, src, bytes) up.put((off), (src), (bytes))
		UP(oPx, mp->proj_x, (size_t)mp->n * 8); UP(oPy, mp->proj_y, (size_t)mp->n * 8); UP(oVc, mp->view_cos, (size_t)mp->n * 8);
		UP(oLv, mp->level, (size_t)mp->n * 4); UP(oPc, mp->cam, (size_t)mp->n * 4); UP(oPd, mp->desc, (size_t)mp->n * mp->stride);
		if (havingMasks) { UP(oPm, mp->mask, (size_t)mp->n * mp->stride); UP(oFm, f->mask, (size_t)f->n * f->stride); }
		UP(oKeys, f->keys, (size_t)f->n * sizeof(mcs_keypoint)); UP(oFd, f->desc, (size_t)f->n * f->stride); UP(oFc, f->cam, (size_t)f->n * 4);
		UP(oAs, f->assigned, (size_t)f->n); UP(oW, f->width, (size_t)f->nr_cams * 4); UP(oH, f->height, (size_t)f->nr_cams * 4);
		UP(oSc, f->scale_factors, (size_t)f->nlevels * 8);
#undef UP
		if (up.flush(s) != hipSuccess) return done(fail(MCS_ERR_HIP, "H2D copy failed"));
		a.px = (const double*)(buf + oPx); a.py = (const double*)(buf + oPy); a.vcos = (const double*)(buf + oVc); a.level = (const int*)(buf + oLv);
		a.pcam = (const int*)(buf + oPc); a.pdesc = buf + oPd; a.pmask = havingMasks ? buf + oPm : nullptr;
		a.keys = (const mcs_keypoint*)(buf + oKeys); a.fdesc = buf + oFd; a.fmask = havingMasks ? buf + oFm : nullptr; a.fcam = (const int*)(buf + oFc);
		a.assigned = buf + oAs; a.width = (const int*)(buf + oW); a.height = (const int*)(buf + oH); a.scales = (const double*)(buf + oSc);
		a.match = (int*)(buf + oMatch); a.nmatches = (int*)(buf + oNm);
	} else {
		a.px = mp->proj_x; a.py = mp->proj_y; a.vcos = mp->view_cos; a.level = mp->level; a.pcam = mp->cam; a.pdesc = mp->desc; a.pmask = mp->mask;
		a.keys = f->keys; a.fdesc = f->desc; a.fmask = f->mask; a.fcam = f->cam; a.assigned = f->assigned; a.width = f->width; a.height = f->height;
		a.scales = f->scale_factors; a.match = match; a.nmatches = nmatches;
	}
	if (mp->n > 0) launch_projection(a, s);
	else if (!host) { HIPCHK(hipMemsetAsync(nmatches, 0, 4, s)); }
	if (hipGetLastError() != hipSuccess) return done(fail(MCS_ERR_HIP, "projection kernels failed to launch"));
	if (host) {
		*nmatches = 0;
		if (mp->n > 0) {
			if (hipMemcpyAsync(match, buf + oMatch, (size_t)mp->n * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
			if (hipMemcpyAsync(nmatches, buf + oNm, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
			if (hipMemcpyAsync(f->assigned, buf + oAs, (size_t)f->n, hipMemcpyDeviceToHost, s) != hipSuccess) return done(fail(MCS_ERR_HIP, "D2H"));
		}
	}
	return done(MCS_OK);   // synchronises: the scratch is freed here (DEVICE kind therefore blocks too; this row is not on the bench path)
}

int mcs_rows_valid(mcs_ctx* c, const int32_t* nkp_dev, int nimg, int cap, uint8_t* valid_dev) {
	if (!c || !nkp_dev || !valid_dev || nimg < 1 || cap < 1) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	launch_rows_valid(nkp_dev, nimg, cap, valid_dev, c->stream);
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

int mcs_rig_pack_headers(mcs_ctx* c, const int32_t* nkp_dev, int nimg, int cap, uint8_t* blocks_dev, int row_stride) {
	if (!c || !nkp_dev || !blocks_dev || nimg < 1 || cap < 1 || row_stride < 4 || (row_stride & 3)) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	launch_rig_pack_headers(nkp_dev, nimg, cap, blocks_dev, row_stride, c->stream);
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

int mcs_rig_rows_valid(mcs_ctx* c, const uint8_t* blocks_dev, int nimg, int cap, int row_stride, uint8_t* valid_dev, int32_t* nkp_out_dev) {
	if (!c || !blocks_dev || !valid_dev || nimg < 1 || cap < 1 || row_stride < 4 || (row_stride & 3)) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	launch_rig_rows_valid(blocks_dev, nimg, cap, row_stride, valid_dev, nkp_out_dev, c->stream);
	HIPCHK(hipGetLastError());
	return MCS_OK;
}

static int single_distance(mcs_ctx* c, const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out) {
	if (!c || !a || !b || !out || (dim != 16 && dim != 32 && dim != 64)) return fail(MCS_ERR_INVALID, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	uint8_t* buf = nullptr;
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(ctx_arena(c, 4 * 64, &buf));   // the context's persistent scratch
	HIPCHK(hipMemcpy(buf, a, dim, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(buf + 64, b, dim, hipMemcpyHostToDevice));
	if (ma) { HIPCHK(hipMemcpy(buf + 128, ma, dim, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(buf + 192, mb, dim, hipMemcpyHostToDevice)); }
	launch_single_distance(buf, buf + 64, ma ? buf + 128 : nullptr, ma ? buf + 192 : nullptr, dim, c->dscalar, c->stream);
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpy(out, c->dscalar, sizeof(int), hipMemcpyDeviceToHost));
	return MCS_OK;
}

int mcs_descriptor_distance(mcs_ctx* c, const uint8_t* a, const uint8_t* b, int dim, int* out) { return single_distance(c, a, b, nullptr, nullptr, dim, out); }
int mcs_descriptor_distance_masked(mcs_ctx* c, const uint8_t* a, const uint8_t* b, const uint8_t* ma, const uint8_t* mb, int dim, int* out) {
	if (!ma || !mb) return fail(MCS_ERR_INVALID, "null masks");
	return single_distance(c, a, b, ma, mb, dim, out);
}

}  // extern "C"

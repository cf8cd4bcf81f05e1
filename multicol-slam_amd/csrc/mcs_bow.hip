// mcs_bow.hip — cMultiFrame::ComputeBoW (src/cMultiFrame.cpp:356-363): the DBoW2 vocabulary-tree descent of every descriptor
// (ThirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259 with FORB::distance, FORB.cpp:85-105), SURVEY §8f row 4.
// One lane per feature: the descriptor (32 bytes, what FORB compares) sits in 8 VGPRs, every level reads the <= k child descriptors of the
// current node (the small vocabulary — 8823 nodes x 32 B — stays in L2) and keeps the first strict minimum, like the reference.
// Outputs per feature: the leaf node reached and the node of the path at level L - levelsup; word ids, weights and the BowVector /
// FeatureVector maps are table look-ups and a sort on the host.
#include "mcs_host.h"

namespace mcs {

struct BowArgs {
	const uint8_t* nodeDesc; const int* childOff; const int* childIdx; int L;
	const uint8_t* desc; int n; int stride; int levelsup;
	int* leaf; int* nid;
};

__global__ __launch_bounds__(256) void k_bow_transform(BowArgs a) {
	const int f = blockIdx.x * 256 + threadIdx.x;
	if (f >= a.n) return;
	uint32_t q[8];
	const uint32_t* qp = reinterpret_cast<const uint32_t*>(a.desc + (size_t)f * a.stride);
#pragma unroll
	for (int w = 0; w < 8; ++w) q[w] = qp[w];
	const int nidLevel = a.L - a.levelsup;
	int nidv = 0, cur = 0, level = 0;
	for (int guard = 0; guard < 64; ++guard) {
		++level;
		const int lo = a.childOff[cur], hi = a.childOff[cur + 1];
		int best = 0x7FFFFFFF, bestId = cur;
		for (int c = lo; c < hi; ++c) {
			const int id = a.childIdx[c];
			const uint4* np = reinterpret_cast<const uint4*>(a.nodeDesc + 32 * (size_t)id);
			const uint4 n0 = np[0], n1 = np[1];
			const int d = __popc(q[0] ^ n0.x) + __popc(q[1] ^ n0.y) + __popc(q[2] ^ n0.z) + __popc(q[3] ^ n0.w) + __popc(q[4] ^ n1.x) +
			              __popc(q[5] ^ n1.y) + __popc(q[6] ^ n1.z) + __popc(q[7] ^ n1.w);
			if (d < best) { best = d; bestId = id; }   // the first child starts the minimum, later ones need strictly less
		}
		cur = bestId;
		if (level == nidLevel) nidv = cur;
		if (a.childOff[cur + 1] - a.childOff[cur] <= 0) break;   // isLeaf()
	}
	a.leaf[f] = cur;
	a.nid[f] = nidv;
}

}  // namespace mcs

using namespace mcs;

struct mcs_vocabulary {
	mcs_ctx* ctx = nullptr;
	int nNodes = 0, L = 0;
	uint8_t* nodeDesc = nullptr; int* childOff = nullptr; int* childIdx = nullptr;
};

int mcs_vocabulary_create(mcs_ctx* c, int n_nodes, const uint8_t* node_desc, const int32_t* child_off, const int32_t* child_idx, int L,
                          mcs_vocabulary** out) {
	if (!c || !node_desc || !child_off || !child_idx || !out) return fail(MCS_ERR_INVALID, "null argument");
	if (n_nodes < 2 || L < 1) return fail(MCS_ERR_INVALID, "a vocabulary needs a root, at least one child and L >= 1");
	// validate the tree: offsets monotone, children in range, the root has children, no self loops (the descent must terminate)
	if (child_off[0] != 0 || child_off[1] <= 0) return fail(MCS_ERR_INVALID, "the root (node 0) must have children");
	for (int i = 0; i < n_nodes; ++i) {
		if (child_off[i + 1] < child_off[i]) return fail(MCS_ERR_INVALID, "child_off must be non-decreasing");
		for (int k = child_off[i]; k < child_off[i + 1]; ++k)
			if (child_idx[k] <= i || child_idx[k] >= n_nodes) return fail(MCS_ERR_INVALID, "children must have larger node ids than their parent (DBoW2 numbers nodes top-down)");
	}
	HIPCHK(hipSetDevice(c->device));
	mcs_vocabulary* v = new mcs_vocabulary();
	v->ctx = c; v->nNodes = n_nodes; v->L = L;
	const size_t nidx = (size_t)child_off[n_nodes];
	bool ok = hipMalloc((void**)&v->nodeDesc, (size_t)n_nodes * 32) == hipSuccess && hipMalloc((void**)&v->childOff, ((size_t)n_nodes + 1) * 4) == hipSuccess &&
	          hipMalloc((void**)&v->childIdx, std::max<size_t>(nidx, 1) * 4) == hipSuccess;
	ok = ok && hipMemcpy(v->nodeDesc, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(v->childOff, child_off, ((size_t)n_nodes + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
	     hipMemcpy(v->childIdx, child_idx, nidx * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) { mcs_vocabulary_destroy(v); return fail(MCS_ERR_HIP, "vocabulary upload failed"); }
	*out = v;
	return MCS_OK;
}

void mcs_vocabulary_destroy(mcs_vocabulary* v) {
	if (!v) return;
	(void)hipFree(v->nodeDesc); (void)hipFree(v->childOff); (void)hipFree(v->childIdx);
	delete v;
}

int mcs_bow_transform(mcs_vocabulary* v, const uint8_t* desc, int n, int stride, int levelsup, mcs_mem_kind kind, int32_t* leaf_node, int32_t* node_at_level) {
	if (!v || !desc || !leaf_node || !node_at_level) return fail(MCS_ERR_INVALID, "null argument");
	if (n < 0 || stride < 32 || (stride & 3)) return fail(MCS_ERR_INVALID, "descriptors must be >= 32 bytes per row (FORB compares 32 bytes), stride a multiple of 4");
	if (n == 0) return MCS_OK;
	mcs_ctx* c = v->ctx;
	HIPCHK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	BowArgs a{v->nodeDesc, v->childOff, v->childIdx, v->L, desc, n, stride, levelsup, leaf_node, node_at_level};
	uint8_t* buf = nullptr;
	const bool host = kind == MCS_MEM_HOST;
	if (host) {
		HIPCHK(ctx_arena(c, (size_t)n * stride + (size_t)n * 8, &buf));
		if (hipMemcpyAsync(buf, desc, (size_t)n * stride, hipMemcpyHostToDevice, s) != hipSuccess) return fail(MCS_ERR_HIP, "H2D copy failed");
		a.desc = buf; a.leaf = (int*)(buf + (size_t)n * stride); a.nid = a.leaf + n;
	}
	hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, s, a);
	hipError_t e = hipGetLastError();
	if (host) {
		if (e == hipSuccess) e = hipMemcpyAsync(leaf_node, a.leaf, (size_t)n * 4, hipMemcpyDeviceToHost, s);
		if (e == hipSuccess) e = hipMemcpyAsync(node_at_level, a.nid, (size_t)n * 4, hipMemcpyDeviceToHost, s);
		const hipError_t e2 = hipStreamSynchronize(s);
		if (e == hipSuccess) e = e2;
	}
	if (e != hipSuccess) return fail(MCS_ERR_HIP, std::string("bow transform: ") + hipGetErrorString(e));
	return MCS_OK;
}

// facade_driver2.cpp — the callers either side of the path through the C++ facade (include/mcs/mcs_facade.hpp): LoadMCS from the reference's
// calibration YAMLs, cORBVocabulary::load / transform (ComputeBoW), WorldToCamHom_fast, SearchByProjection(F, mapPoints), the best-in-window loop of
// Fuse / SearchBySim3, ComputeDistinctiveDescriptors and the vocabulary-restricted SearchByBoW(KF, F).  tests/test_gpu_cpp_facade.py compares
// every output array with the oracle.
//   argv[1] = directory with MultiCamSys_Calibration.yaml, InteriorOrientationFisheye<c>.yaml, voc.yml, frames.bin (2 x ncam images w*h)
//   argv[2] = output file
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include "mcs/mcs_facade.hpp"

using namespace MultiColSLAM;

template <class T> static void wr(FILE* f, const T* p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) std::exit(2); }
template <class T> static void wrv(FILE* f, const std::vector<T>& v) { const int32_t n = (int32_t)v.size(); wr(f, &n, 1); wr(f, v.data(), v.size()); }

int main(int argc, char** argv) {
	if (argc != 3) return 1;
	try {
		const std::string dir = argv[1];
		FILE* fo = std::fopen(argv[2], "wb");
		if (!fo) return 1;
		Context ctx(0);
		cMultiCamSys_ rig;
		LoadMCS(dir, rig);
		const int ncam = rig.GetNrCams(), w = rig.camModels[0].ocam.width, h = rig.camModels[0].ocam.height;
		for (int c = 0; c < ncam; ++c) {
			wr(fo, &rig.camModels[c].ocam, 1); wr(fo, rig.M_c[c].data(), 16);
			int64_t sum = 0; for (size_t i = 0; i < (size_t)w * h; ++i) sum += rig.camModels[c].mirrorMask0.data[i];
			wr(fo, &sum, 1);
		}
		// a small rig motion, so that MtMc_inv is not the calibration alone
		Matx44d Mt{}; Mt[0] = Mt[5] = Mt[10] = Mt[15] = 1.0; Mt[3] = 0.02; Mt[7] = -0.01; Mt[11] = 0.03; Mt[1] = 1e-3; Mt[4] = -1e-3;
		rig.Set_M_t(Mt);
		for (int c = 0; c < ncam; ++c) wr(fo, rig.MtMc_inv[c].data(), 16);

		std::vector<std::vector<uint8_t>> imgs(2 * ncam, std::vector<uint8_t>((size_t)w * h));
		{
			FILE* fi = std::fopen((dir + "/frames.bin").c_str(), "rb");
			if (!fi) return 1;
			for (auto& v : imgs) if (fread(v.data(), 1, v.size(), fi) != v.size()) return 2;
			std::fclose(fi);
		}
		mdBRIEFextractorOct ex(ctx, 600, 1.2f, 8, 25, 0, mdBRIEFextractorOct::HARRIS_SCORE, 32, 20, false, 2, true, true, 32);
		struct Flat { std::vector<KeyPoint> keys; std::vector<uint8_t> d, m; std::vector<Vec3d> rays; std::vector<int32_t> cam; } fr[2];
		std::vector<const uint8_t*> mp; std::vector<mcs_ocam> oc;
		for (auto& m : rig.camModels) { mp.push_back(m.mirrorMask0.data); oc.push_back(m.ocam); }
		for (int f = 0; f < 2; ++f) {
			std::vector<const uint8_t*> ip;
			for (int c = 0; c < ncam; ++c) ip.push_back(imgs[f * ncam + c].data());
			std::vector<std::vector<KeyPoint>> keys; std::vector<Mat8u> desc, dmask; std::vector<std::vector<Vec3d>> rays;
			ex.extractRig(ip, w, h, w, mp, oc, keys, desc, dmask, rays);
			for (int c = 0; c < ncam; ++c) {
				const size_t n = keys[c].size();
				fr[f].keys.insert(fr[f].keys.end(), keys[c].begin(), keys[c].end());
				fr[f].rays.insert(fr[f].rays.end(), rays[c].begin(), rays[c].end());
				if (n) { fr[f].d.insert(fr[f].d.end(), desc[c].data, desc[c].data + n * 32); fr[f].m.insert(fr[f].m.end(), dmask[c].data, dmask[c].data + n * 32); }
				fr[f].cam.insert(fr[f].cam.end(), n, c);
			}
			wrv(fo, fr[f].cam); wr(fo, fr[f].keys.data(), fr[f].keys.size()); wr(fo, fr[f].d.data(), fr[f].d.size()); wr(fo, fr[f].m.data(), fr[f].m.size());
		}
		const int n0 = (int)fr[0].keys.size(), n1 = (int)fr[1].keys.size();

		// ComputeBoW of both frames
		cORBVocabulary voc(ctx);
		voc.load(dir + "/voc.yml");
		std::vector<int32_t> node[2];
		for (int f = 0; f < 2; ++f) {
			cORBVocabulary::BowVector bv; cORBVocabulary::FeatureVector fv;
			voc.transform(fr[f].d.data(), (int)fr[f].keys.size(), 32, bv, fv, 4);
			node[f].assign(fr[f].keys.size(), -1);
			for (auto& e : fv) for (unsigned i : e.second) node[f][i] = (int32_t)e.first;
			wrv(fo, node[f]);
			std::vector<int32_t> ids; std::vector<double> vals;
			for (auto& e : bv) { ids.push_back((int32_t)e.first); vals.push_back(e.second); }
			wrv(fo, ids); wrv(fo, vals);
		}

		// WorldToCamHom_fast: the points 2.5 units along every bearing ray of frame 0, moved to the world by MtMc[c]
		std::vector<double> pts(3 * (size_t)n0), uv(2 * (size_t)n0);
		std::vector<uint8_t> fl(n0);
		for (int i = 0; i < n0; ++i) {
			const Matx44d& M = rig.MtMc[fr[0].cam[i]];
			const double p[4] = {fr[0].rays[i].v[0] * 2.5, fr[0].rays[i].v[1] * 2.5, fr[0].rays[i].v[2] * 2.5, 1.0};
			for (int r = 0; r < 3; ++r) { double s = 0; for (int k = 0; k < 4; ++k) s += M[4 * r + k] * p[k]; pts[3 * i + r] = s; }
		}
		rig.WorldToCamHom_fast(ctx, pts.data(), fr[0].cam.data(), n0, uv.data(), fl.data());
		wr(fo, pts.data(), pts.size()); wr(fo, uv.data(), uv.size()); wr(fo, fl.data(), fl.size());

		// frame 1 as the grid frame
		cORBmatcher matcher(ctx, 0.8, false, 32, true);
		cORBmatcher::FrameGridView G;
		G.mvKeys = fr[1].keys.data(); G.descriptors = fr[1].d.data(); G.masks = fr[1].m.data(); G.keypoint_to_cam = fr[1].cam; G.n = n1;
		G.hasMapPoint.assign(n1, 0);
		for (int i = 0; i < n1; i += 9) G.hasMapPoint[i] = 1;
		for (int c = 0; c < ncam; ++c) { G.width.push_back(w); G.height.push_back(h); }
		double s = 1.0;
		for (int l = 0; l < 8; ++l) { G.mvScaleFactors.push_back(s); s *= ex.GetScaleFactor(); }
		// SearchByProjection(F, mapPoints, th = 3): the features of frame 0 "projected" 3 px right, 1 px down
		cORBmatcher::Projections P;
		for (int i = 0; i < n0; ++i) {
			P.x.push_back(fr[0].keys[i].ptx + 3.0); P.y.push_back(fr[0].keys[i].pty + 1.0); P.viewCos.push_back(i % 3 ? 0.9995 : 0.9);
			P.level.push_back(fr[0].keys[i].octave); P.cam.push_back(fr[0].cam[i]);
		}
		P.desc = fr[0].d.data(); P.mask = fr[0].m.data();
		std::vector<int> m;
		const std::vector<uint8_t> pre = G.hasMapPoint;
		int32_t nm = matcher.SearchByProjection(G, P, 3.0, m);
		wr(fo, &nm, 1); wrv(fo, m); wr(fo, pre.data(), pre.size()); wr(fo, G.hasMapPoint.data(), G.hasMapPoint.size());
		// best-in-window loop (Fuse / SearchBySim3): radius 8 x scale, levels octave-1 .. octave, accept <= TH_LOW = 32
		cORBmatcher::Windows W;
		for (int i = 0; i < n0; ++i) {
			const int o = fr[0].keys[i].octave;
			W.x.push_back(P.x[i]); W.y.push_back(P.y[i]); W.r.push_back(8.0 * G.mvScaleFactors[o]); W.lo.push_back(o - 1); W.hi.push_back(o); W.cam.push_back(fr[0].cam[i]);
		}
		W.desc = fr[0].d.data(); W.mask = fr[0].m.data();
		std::vector<int> bm, bd;
		nm = matcher.BestInWindows(G, W, 32, false, bm, bd);
		wr(fo, &nm, 1); wrv(fo, bm); wrv(fo, bd);
		// ComputeDistinctiveDescriptors: "map points" observed by 1..7 consecutive rows of frame 0
		std::vector<int32_t> off(1, 0), best;
		for (int k = 0; off.back() < n0; ++k) off.push_back(std::min(n0, off.back() + 1 + k % 7));
		ComputeDistinctiveDescriptors(ctx, fr[0].d.data(), fr[0].m.data(), 32, 32, off, best);
		wrv(fo, off); wrv(fo, best);
		// vocabulary-restricted SearchByBoW(KF = frame 0, F = frame 1)
		std::vector<std::pair<int32_t, int32_t>> order;   // (node, index)
		for (int i = 0; i < n0; ++i) if (node[0][i] >= 0) order.emplace_back(node[0][i], i);
		std::sort(order.begin(), order.end());
		std::vector<uint8_t> kd(order.size() * 32), km(order.size() * 32);
		FeatureSetView kf, F;
		for (size_t r = 0; r < order.size(); ++r) {
			std::memcpy(&kd[r * 32], &fr[0].d[(size_t)order[r].second * 32], 32); std::memcpy(&km[r * 32], &fr[0].m[(size_t)order[r].second * 32], 32);
			kf.flag.push_back(order[r].second % 5 != 0);   // "has a good map point"
			kf.cam.push_back(order[r].first);
		}
		kf.descriptors = kd.data(); kf.masks = km.data(); kf.n = (int)order.size();
		F.descriptors = fr[1].d.data(); F.masks = fr[1].m.data(); F.n = n1; F.cam = node[1];
		for (int j = 0; j < n1; ++j) F.flag.push_back(node[1][j] >= 0);
		std::vector<int> mF;
		nm = matcher.SearchByBoWFrameVocabulary(kf, F, mF);
		for (auto& v : mF) if (v >= 0) v = order[v].second;   // back to frame-0 feature indices
		wr(fo, &nm, 1); wrv(fo, mF);
		std::fclose(fo);
		return 0;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "facade_driver2: %s\n", e.what());
		return 3;
	}
}

// facade_driver.cpp — exercises include/mcs/mcs_facade.hpp (the C++ host side a MultiCol-SLAM build would link) end to end:
// two 3-camera multi-frames -> mdBRIEFextractorOct::extractRig -> cORBmatcher::SearchByBoW / WindowSearch / SearchForInitialization.
// Input and output are flat binary files so that tests/test_gpu_cpp_facade.py can compare every array with the oracle.
//   in : int32 ncam, w, h, nframes | ncam x mcs_ocam | nframes*ncam images (w*h) | ncam masks (w*h)
//   out: per frame: int32 n[ncam]; keypoints (28 B), descriptors (32 B), masks (32 B), rays (24 B) of all cameras concatenated |
//        int32 nBoW, match12[n0] | int32 nWin, vnMatches21[n1] | int32 nIni, vnMatches12[n0], prevMatched[2*n0] doubles
#include <cstdio>
#include <cstdlib>
#include "mcs/mcs_facade.hpp"

using namespace MultiColSLAM;

template <class T> static void rd(FILE* f, T* p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }
template <class T> static void wr(FILE* f, const T* p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) std::exit(2); }

struct Flat { std::vector<KeyPoint> keys; std::vector<uint8_t> d, m; std::vector<Vec3d> rays; std::vector<int32_t> cam; };

int main(int argc, char** argv) {
	if (argc != 3) return 1;
	try {
		FILE* fi = std::fopen(argv[1], "rb");
		FILE* fo = std::fopen(argv[2], "wb");
		if (!fi || !fo) return 1;
		int32_t hdr[4];
		rd(fi, hdr, 4);
		const int ncam = hdr[0], w = hdr[1], h = hdr[2], nframes = hdr[3];
		std::vector<mcs_ocam> cams(ncam);
		rd(fi, cams.data(), ncam);
		std::vector<std::vector<uint8_t>> imgs(nframes * ncam, std::vector<uint8_t>((size_t)w * h)), masks(ncam, std::vector<uint8_t>((size_t)w * h));
		for (auto& v : imgs) rd(fi, v.data(), v.size());
		for (auto& v : masks) rd(fi, v.data(), v.size());
		Context ctx(0);
		mdBRIEFextractorOct ex(ctx, 1000, 1.2f, 8, 25, 0, mdBRIEFextractorOct::HARRIS_SCORE, 32, 20, false, 2, true, true, 32);
		std::vector<Flat> fr(nframes);
		std::vector<const uint8_t*> mp;
		for (auto& v : masks) mp.push_back(v.data());
		for (int f = 0; f < nframes; ++f) {
			std::vector<const uint8_t*> ip;
			for (int c = 0; c < ncam; ++c) ip.push_back(imgs[f * ncam + c].data());
			std::vector<std::vector<KeyPoint>> keys; std::vector<Mat8u> desc, dmask; std::vector<std::vector<Vec3d>> rays;
			ex.extractRig(ip, w, h, w, mp, cams, keys, desc, dmask, rays);
			std::vector<int32_t> n(ncam);
			Flat& F = fr[f];
			for (int c = 0; c < ncam; ++c) {
				n[c] = (int32_t)keys[c].size();
				F.keys.insert(F.keys.end(), keys[c].begin(), keys[c].end());
				F.rays.insert(F.rays.end(), rays[c].begin(), rays[c].end());
				if (n[c]) { F.d.insert(F.d.end(), desc[c].data, desc[c].data + (size_t)n[c] * 32); F.m.insert(F.m.end(), dmask[c].data, dmask[c].data + (size_t)n[c] * 32); }
				F.cam.insert(F.cam.end(), n[c], c);
			}
			wr(fo, n.data(), ncam); wr(fo, F.keys.data(), F.keys.size()); wr(fo, F.d.data(), F.d.size()); wr(fo, F.m.data(), F.m.size());
			wr(fo, F.rays.data(), F.rays.size());
		}
		const int n0 = (int)fr[0].keys.size(), n1 = (int)fr[1].keys.size();
		cORBmatcher matcher(ctx, 0.8, false, 32, true);
		FeatureSetView a, b;
		a.descriptors = fr[0].d.data(); a.masks = fr[0].m.data(); a.flag.assign(n0, 1); a.n = n0;
		b.descriptors = fr[1].d.data(); b.masks = fr[1].m.data(); b.flag.assign(n1, 1); b.n = n1;
		std::vector<int> m12;
		int32_t nb = matcher.SearchByBoW(a, b, m12);
		wr(fo, &nb, 1); wr(fo, m12.data(), m12.size());
		cORBmatcher::FrameGridView G[2];
		for (int f = 0; f < 2; ++f) {
			G[f].mvKeys = fr[f].keys.data(); G[f].descriptors = fr[f].d.data(); G[f].masks = fr[f].m.data(); G[f].keypoint_to_cam = fr[f].cam;
			G[f].hasMapPoint.assign(fr[f].keys.size(), 0); G[f].n = (int)fr[f].keys.size();
			for (int c = 0; c < ncam; ++c) { G[f].width.push_back(cams[c].width); G[f].height.push_back(cams[c].height); }
			double s = 1.0;
			for (int l = 0; l < 8; ++l) { G[f].mvScaleFactors.push_back(s); s *= ex.GetScaleFactor(); }
		}
		std::vector<uint8_t> good(n0, 1);
		std::vector<int> m21;
		int32_t nw = matcher.WindowSearch(G[0], good, G[1], 60, m21, 2);
		wr(fo, &nw, 1); wr(fo, m21.data(), m21.size());
		std::vector<double> prev(2 * (size_t)n0);
		for (int i = 0; i < n0; ++i) { prev[2 * i] = fr[0].keys[i].ptx; prev[2 * i + 1] = fr[0].keys[i].pty; }
		std::vector<int> i12;
		cORBmatcher ini(ctx, 0.9, false, 32, true);
		int32_t ni = ini.SearchForInitialization(G[0], G[1], prev, i12, 50);
		wr(fo, &ni, 1); wr(fo, i12.data(), i12.size()); wr(fo, prev.data(), prev.size());
		std::fclose(fo);
		return 0;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "facade_driver: %s\n", e.what());
		return 3;
	}
}

/*
 * srba-slam -- command-line front-end over the MI355X back-end, mirroring the reference's apps/srba-slam:
 *   flags            apps/srba-slam/srba-slam_main.cpp:66-104 (the GUI / video ones are accepted and ignored)
 *   dataset formats  apps/srba-slam/CDatasetParserBase.h:56-112 (text matrix "FRAME_ID FEAT_ID fields...", '%' / '#' comments),
 *                    CDatasetParser_RelGraphSLAM2D.h:28-52 (12 columns), _RangeBearing2D.h:30-52 (4), _Cartesian_3D.h:28-47 (5),
 *                    _Monocular.h:28-46 (4), _Stereo.h:28-48 (6); ground-truth path "idx x y z qr qx qy qz" (CDatasetParserBase.h:213-226)
 *   run loop         apps/srba-slam/srba-run-generic-impl.h:105-182 (parameters), :337-470 (one key-frame per FRAME_ID; graph-SLAM adds the
 *                    fixed self-landmark), :512 (per key-frame RMSE), eval_overall_squared_error at the end
 *   problem types    instance_relative_graph_slam_se2.cpp, instance_se2_lm2d_rangebearing2d.cpp, instance_se3_lm3d_cartesian3d.cpp,
 *                    instance_se3_lm3d_monocular.cpp, instance_se3_lm3d_stereo.cpp  (same RBA_OPTIONS per type)
 * Numeric back-end: the GPU (libsrba_hip). There is no CPU fallback.
 */
#include <srba.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <random>
#include <set>

using namespace srba;

namespace {

struct Args {
	std::map<std::string, std::string> val; std::set<std::string> flag;
	bool has(const std::string &k) const { return val.count(k) || flag.count(k); }
	std::string str(const std::string &k, const std::string &def = "") const { std::map<std::string, std::string>::const_iterator it = val.find(k); return it == val.end() ? def : it->second; }
	double num(const std::string &k, double def) const { return val.count(k) ? std::atof(val.find(k)->second.c_str()) : def; }
};
const char *kValueArgs[] = {"dataset", "gt-map", "gt-path", "max-fixed-feats-per-kf", "obs", "sensor-params-cfg-file", "profile-stats", "profile-stats-length", "noise", "noise-ang",
	"max-spanning-tree-depth",
	"max-optimize-depth", "max-lambda", "max-iters", "submap-size", "verbose", "random-seed", "cfg-file-rba", "cfg-file-rba-bootstrap", "create-video", "gui-delay", "video-fps", "save-final-graph",
	"save-final-graph-landmarks", "device", "save-edges", NULL};
const char *kFlagArgs[] = {"se2", "se3", "lm-2d", "lm-3d", "graph-slam", "list-problems", "no-gui", "step-by-step", "add-noise", "debug-dump-cur-spantree", "eval-overall-sqr-error",
	"eval-overall-se3-error",
	"eval-connectivity", "parse-only", "help", NULL};

Args parse_args(int argc, char **argv) {
	Args a;
	for (int i = 1; i < argc; i++) {
		std::string k = argv[i];
		if (k == "-d") k = "--dataset"; else if (k == "-v") k = "--verbose"; else if (k == "-h") k = "--help";
		if (k.size() < 3 || k.substr(0, 2) != "--") throw std::runtime_error("unexpected argument: " + k);
		k = k.substr(2);
		bool known = false;
		for (int j = 0; kFlagArgs[j]; j++) if (k == kFlagArgs[j]) { a.flag.insert(k); known = true; }
		for (int j = 0; kValueArgs[j] && !known; j++) if (k == kValueArgs[j]) { if (i + 1 >= argc) throw std::runtime_error("missing value for --" + k); a.val[k] = argv[++i]; known = true; }
		if (!known) throw std::runtime_error("unknown argument: --" + k + " (see --help)");
	}
	return a;
}

/** CDatasetParserBase::load_obs: a dense text matrix, one observation per row */
struct Dataset {
	std::vector<std::vector<double> > rows; size_t cols = 0;
	void load(const std::string &file) {
		std::ifstream f(file.c_str()); if (!f) throw std::runtime_error("cannot open dataset file: " + file);
		std::string line;
		while (std::getline(f, line)) {
			const size_t p = line.find_first_not_of(" \t\r"); if (p == std::string::npos || line[p] == '%' || line[p] == '#') continue;
			std::istringstream ss(line); std::vector<double> r; double v; while (ss >> v) r.push_back(v);
			if (r.empty()) continue;
			if (!cols) cols = r.size(); else if (r.size() != cols) throw std::runtime_error("dataset: rows of different length");
			rows.push_back(r);
		}
		if (rows.size() <= 2) throw std::runtime_error("dataset: too few observations"); // ASSERT_ABOVE_(getRowCount,2)
	}
};
/** minimal INI reader for --sensor-params-cfg-file ([EXT] TStereoCamera::loadFromConfigFile("CAMERA"): sections CAMERA_LEFT, CAMERA_RIGHT, CAMERA_LEFT2RIGHT_POSE) */
struct Ini {
	std::map<std::string, std::map<std::string, std::string> > sec;
	void load(const std::string &file) {
		std::ifstream f(file.c_str()); if (!f) throw std::runtime_error("cannot open config file: " + file);
		std::string line, cur;
		while (std::getline(f, line)) {
			const size_t c = line.find_first_of(";#"); if (c != std::string::npos && (c == 0 || line[c - 1] != '[')) line = line.substr(0, c == std::string::npos ? line.size() : (line.substr(0,
				c).find('=') == std::string::npos ? 0 : c));
			const size_t a = line.find('['), b = line.find(']');
			if (a != std::string::npos && b != std::string::npos && line.find('=') == std::string::npos) { cur = line.substr(a + 1, b - a - 1); continue; }
			const size_t e = line.find('='); if (e == std::string::npos) continue;
			std::string k = line.substr(0, e), v = line.substr(e + 1);
			k.erase(0, k.find_first_not_of(" \t")); k.erase(k.find_last_not_of(" \t") + 1); v.erase(0, v.find_first_not_of(" \t")); v.erase(v.find_last_not_of(" \t\r") + 1);
			sec[cur][k] = v;
		}
	}
	double num(const std::string &s, const std::string &k) const {
		std::map<std::string, std::map<std::string, std::string> >::const_iterator it = sec.find(s); if (it == sec.end() || !it->second.count(k)) throw std::runtime_error("config file: missing [" + s
			+ "] " + k);
		return std::atof(it->second.find(k)->second.c_str());
	}
	std::vector<double> vec(const std::string &s, const std::string &k) const {
		std::map<std::string, std::map<std::string, std::string> >::const_iterator it = sec.find(s); if (it == sec.end() || !it->second.count(k)) throw std::runtime_error("config file: missing [" + s
			+ "] " + k);
		std::string v = it->second.find(k)->second; for (size_t i = 0; i < v.size(); i++) if (v[i] == '[' || v[i] == ']' || v[i] == ',') v[i] = ' ';
		std::istringstream ss(v); std::vector<double> r; double x; while (ss >> x) r.push_back(x); return r;
	}
};
void load_camera(const Ini &ini, const std::string &section, mrpt::utils::TCamera &c) { c.fx(ini.num(section, "fx")); c.fy(ini.num(section, "fy")); c.cx(ini.num(section, "cx")); c.cy(ini.num(section,
	"cy")); }

std::mt19937_64 g_rng(0);
double gauss(double sigma) { std::normal_distribution<double> d(0.0, sigma); return d(g_rng); }

// ---- per observation type: column count, row -> observation, noise / sensor parameters (CDatasetParser_*.h, instance_*.cpp) ----
template <class OBS> struct Parser;
template <> struct Parser<observations::RelativePoses_2D> {
	static const size_t COLS = 12; double sxy, syaw; // KeyframeIndex LandmarkID | X Y Z YAW PITCH ROLL QR QX QY QZ
	explicit Parser(const Args &a) : sxy(a.num("noise", 0.10)), syaw((a.has("noise-ang") ? a.num("noise-ang", 4.0) : 4.0) * M_PI / 180.0) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.x = r[2] + (noisy ? gauss(sxy) : 0); o.obs_data.y = r[3] + (noisy ? gauss(sxy) : 0);
		o.obs_data.yaw = r[5] + (noisy ? gauss(syaw) : 0); }
	template <class RBA> void params(RBA &rba, const Args &) const { rba.parameters.obs_noise.lambda.setZero(); rba.parameters.obs_noise.lambda(0, 0) = rba.parameters.obs_noise.lambda(1,
		1) = 1.0 / (sxy * sxy); rba.parameters.obs_noise.lambda(2, 2) = 1.0 / (syaw * syaw); }
};
template <> struct Parser<observations::RangeBearing_2D> {
	static const size_t COLS = 4; double sr, sy;
	explicit Parser(const Args &a) : sr(a.num("noise", 1e-4)), sy(a.has("noise") ? a.num("noise", 1e-5) : 1e-5) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.range = r[2] + (noisy ? gauss(sr) : 0); o.obs_data.yaw = r[3] + (noisy ? gauss(sy) : 0); }
	template <class RBA> void params(RBA &rba, const Args &) const { rba.parameters.obs_noise.std_noise_observations = sr; }
};
template <> struct Parser<observations::Cartesian_2D> {
	static const size_t COLS = 4; double s;
	explicit Parser(const Args &a) : s(a.num("noise", 1e-3)) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.pt.x = r[2] + (noisy ? gauss(s) : 0); o.obs_data.pt.y = r[3] + (noisy ? gauss(s) : 0); }
	template <class RBA> void params(RBA &rba, const Args &) const { rba.parameters.obs_noise.std_noise_observations = s; }
};
template <> struct Parser<observations::Cartesian_3D> {
	static const size_t COLS = 5; double s;
	explicit Parser(const Args &a) : s(a.num("noise", 1e-3)) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.pt.x = r[2] + (noisy ? gauss(s) : 0); o.obs_data.pt.y = r[3] + (noisy ? gauss(s) : 0);
		o.obs_data.pt.z = r[4] + (noisy ? gauss(s) : 0); }
	template <class RBA> void params(RBA &rba, const Args &) const { rba.parameters.obs_noise.std_noise_observations = s; }
};
template <> struct Parser<observations::RangeBearing_3D> { // CDatasetParser_RangeBearing3D: FRAME_ID FEAT_ID range yaw pitch
	static const size_t COLS = 5; double sr, sa;
	explicit Parser(const Args &a) : sr(a.num("noise", 1e-4)), sa(a.has("noise") ? a.num("noise", 1e-5) : 1e-5) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.range = r[2] + (noisy ? gauss(sr) : 0); o.obs_data.yaw = r[3] + (noisy ? gauss(sa) : 0);
		o.obs_data.pitch = r[4] + (noisy ? gauss(sa) : 0); }
	template <class RBA> void params(RBA &rba, const Args &) const { rba.parameters.obs_noise.std_noise_observations = sr; }
};
template <> struct Parser<observations::MonocularCamera> {
	static const size_t COLS = 4; double s;
	explicit Parser(const Args &a) : s(a.num("noise", 1e-4)) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.px.x = r[2] + (noisy ? gauss(s) : 0); o.obs_data.px.y = r[3] + (noisy ? gauss(s) : 0); }
	template <class RBA> void params(RBA &rba, const Args &a) const {
		rba.parameters.obs_noise.std_noise_observations = s;
		if (!a.has("sensor-params-cfg-file")) throw std::runtime_error("Error: --sensor-params-cfg-file is mandatory for this type of observations.");
		Ini ini; ini.load(a.str("sensor-params-cfg-file")); load_camera(ini, "CAMERA", rba.parameters.sensor.camera_calib);
		rba.parameters.sensor_pose.relative_pose = mrpt::poses::CPose3D(0, 0, 0, -M_PI / 2, 0, -M_PI / 2);
	}
};
template <> struct Parser<observations::StereoCamera> {
	static const size_t COLS = 6; double s;
	explicit Parser(const Args &a) : s(a.num("noise", 1e-4)) {}
	template <class O> void get(const std::vector<double> &r, O &o, bool noisy) const { o.obs_data.left_px.x = r[2] + (noisy ? gauss(s) : 0); o.obs_data.left_px.y = r[3] + (noisy ? gauss(s) : 0);
		o.obs_data.right_px.x = r[4] + (noisy ? gauss(s) : 0); o.obs_data.right_px.y = r[5] + (noisy ? gauss(s) : 0); }
	template <class RBA> void params(RBA &rba, const Args &a) const {
		rba.parameters.obs_noise.std_noise_observations = s;
		if (!a.has("sensor-params-cfg-file")) throw std::runtime_error("Error: --sensor-params-cfg-file is mandatory for this type of observations.");
		Ini ini; ini.load(a.str("sensor-params-cfg-file"));
		load_camera(ini, "CAMERA_LEFT", rba.parameters.sensor.camera_calib.leftCamera); load_camera(ini, "CAMERA_RIGHT", rba.parameters.sensor.camera_calib.rightCamera);
		const std::vector<double> q = ini.vec("CAMERA_LEFT2RIGHT_POSE", "pose_quaternion"); if (q.size() != 7) throw std::runtime_error("config file: pose_quaternion needs 7 numbers");
		rba.parameters.sensor.camera_calib.rightCameraPose = mrpt::poses::CPose3DQuat(q[0], q[1], q[2], mrpt::math::CQuaternionDouble(q[3], q[4], q[5], q[6]));
		if (q[0] == 0) throw std::runtime_error("stereo baseline is zero");
		rba.parameters.sensor_pose.relative_pose = mrpt::poses::CPose3D(0, 0, 0, -M_PI / 2, 0, -M_PI / 2);
	}
};

struct OPT_GRAPH_SLAM : public RBA_OPTIONS_DEFAULT { typedef options::observation_noise_constant_matrix<observations::RelativePoses_2D> obs_noise_matrix_t;
	typedef options::solver_LM_no_schur_sparse_cholesky solver_t; };
struct OPT_CAMERA : public RBA_OPTIONS_DEFAULT { typedef options::sensor_pose_on_robot_se3 sensor_pose_on_robot_t; typedef options::observation_noise_identity obs_noise_matrix_t;
	typedef options::solver_LM_schur_dense_cholesky solver_t; };

template <class KF, class LM, class OBS, class OPT>
int run(const Args &a, const Dataset &ds) {
	typedef RbaEngine<KF, LM, OBS, OPT> my_srba_t;
	const Parser<OBS> parser(a);
	if (ds.cols != Parser<OBS>::COLS) { std::ostringstream m; m << "dataset has " << ds.cols << " columns, this observation type needs " << Parser<OBS>::COLS; throw std::runtime_error(m.str()); }
	const int verbose = (int)a.num("verbose", 1); const bool graph_slam = a.has("graph-slam"), noisy = a.has("add-noise");
	if (a.has("parse-only")) { std::cout << "Loaded " << ds.rows.size() << " observations, " << (size_t)(ds.rows.back()[0] + 1) << " key-frames.\n"; return 0; }
	my_srba_t rba;
	if (a.has("device")) rba.set_hip_device((int)a.num("device", -1));
	parser.params(rba, a);
	rba.setVerbosityLevel(verbose);
	rba.parameters.srba.use_robust_kernel = false; rba.parameters.srba.max_error_per_obs_to_stop = 1e-8; // srba-run-generic-impl.h:127-131
	if (a.has("max-spanning-tree-depth")) rba.parameters.srba.max_tree_depth = (size_t)a.num("max-spanning-tree-depth", 4);
	if (a.has("max-optimize-depth")) rba.parameters.srba.max_optimize_depth = (size_t)a.num("max-optimize-depth", 4);
	if (a.has("max-lambda")) rba.parameters.srba.max_lambda = a.num("max-lambda", 1e20);
	if (a.has("max-iters")) rba.parameters.srba.max_iters = (size_t)a.num("max-iters", 20);
	if (a.has("submap-size")) rba.parameters.ecp.submap_size = (size_t)a.num("submap-size", 20);
	if (graph_slam) { rba.parameters.ecp.min_obs_to_loop_closure = 1; rba.parameters.srba.optimize_new_edges_alone = true; }
	const long seed = (long)a.num("random-seed", -1); g_rng.seed(seed < 0 ? std::random_device()() : (unsigned long)seed);

	const size_t nTotalObs = ds.rows.size(); size_t obsIdx = 0; TKeyFrameID next_kf = 0; double sum_rmse = 0; size_t n_trials = 0, n_kfs = 0;
	const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	typename my_srba_t::TNewKeyFrameInfo info;
	while (obsIdx < nTotalObs) {
		typename my_srba_t::new_kf_observations_t list;
		if (graph_slam) { typename my_srba_t::new_kf_observation_t f; f.is_fixed = true; f.is_unknown_with_init_val = false; f.obs.feat_id = next_kf; list.push_back(f); }
			// the fixed "fake landmark" = the key-frame itself
		while (obsIdx < nTotalObs && (TKeyFrameID)ds.rows[obsIdx][0] == next_kf) {
			typename my_srba_t::new_kf_observation_t o; o.is_fixed = false; o.is_unknown_with_init_val = false; o.obs.feat_id = (TLandmarkID)ds.rows[obsIdx][1];
			parser.get(ds.rows[obsIdx], o.obs, noisy); list.push_back(o); obsIdx++;
		}
		if (list.empty()) throw std::runtime_error("dataset: key-frame ids must be consecutive and start at 0");
		rba.define_new_keyframe(list, info, true);
		const double rmse = info.optimize_results.num_observations ? std::sqrt(info.optimize_results.total_sqr_error_final / info.optimize_results.num_observations) : 0;
		sum_rmse += rmse; n_trials += info.optimize_results.lm.num_trials; n_kfs++;
		if (verbose >= 2) std::printf("KF %6lu: %2lu new edges, %4lu obs, %3lu k2k unknowns, RMSE %.6g -> %.6g\n", (unsigned long)info.kf_id, (unsigned long)info.created_edge_ids.size(),
			(unsigned long)info.optimize_results.num_observations,
			(unsigned long)info.optimize_results.num_kf2kf_edges_optimized, std::sqrt(info.optimize_results.total_sqr_error_init / std::max<size_t>(1, info.optimize_results.num_observations)), rmse);
		next_kf = info.kf_id + 1;
	}
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
	if (verbose >= 1) std::printf("Processed %lu key-frames, %lu observations in %.3f s (%.3f ms/KF, %lu LM iterations); mean per-KF RMSE %.6g; %lu kf2kf edges\n", (unsigned long)n_kfs,
		(unsigned long)nTotalObs, dt, 1e3 * dt / std::max<size_t>(1, n_kfs),
		(unsigned long)n_trials, sum_rmse / std::max<size_t>(1, n_kfs), (unsigned long)rba.get_k2k_edges().size());
	if (a.has("eval-overall-sqr-error")) { const double e = rba.eval_overall_squared_error(); std::printf("eval_overall_squared_error: %.10g\n", e); }
	if (a.has("save-edges")) { // extension: "id from to inv_pose..." one edge per line
		std::ofstream f(a.str("save-edges").c_str()); f.precision(17);
		for (size_t i = 0; i < rba.get_k2k_edges().size(); i++) { const typename my_srba_t::k2k_edge_t &e = rba.get_k2k_edges()[i]; double p[12]; e.inv_pose.storeTo(p);
			f << e.id << " " << e.from << " " << e.to; for (size_t k = 0; k < my_srba_t::pose_t::storage_doubles(); k++) f << " " << p[k]; f << "\n"; }
	}
	if (a.has("save-final-graph")) { if (!rba.save_graph_as_dot(a.str("save-final-graph"), false)) throw std::runtime_error("cannot write " + a.str("save-final-graph")); }   // key-frames only
	if (a.has("save-final-graph-landmarks")) { if (!rba.save_graph_as_dot(a.str("save-final-graph-landmarks"), true)) throw std::runtime_error("cannot write " + a.str("save-final-graph-landmarks")); }
	return 0;
}

void list_problems() {
	std::cout << "Implemented RBA problem types:\n"
		" --se2 --graph-slam\n --se2 --lm-2d --obs RangeBearing_2D\n --se2 --lm-2d --obs Cartesian_2D\n --se3 --lm-3d --obs Cartesian_3D\n"
		" --se3 --lm-3d --obs RangeBearing_3D\n --se3 --lm-3d --obs MonocularCamera\n --se3 --lm-3d --obs StereoCamera\n";
}

} // namespace

int main(int argc, char **argv) {
	try {
		const Args a = parse_args(argc, argv);
		if (a.has("help")) { std::cout << "srba-slam (MI355X back-end). Arguments:"; for (int j = 0; kValueArgs[j]; j++) std::cout << " --" << kValueArgs[j] << " <v>"; for (int j = 0; kFlagArgs[j];
			j++) std::cout << " --" << kFlagArgs[j]; std::cout << "\n"; return 0; }
		if (a.has("list-problems")) { list_problems(); return 0; }
		if (!a.has("obs") && !a.has("graph-slam")) throw std::runtime_error("Error: argument --obs is mandatory (in non-graph-SLAM) to select the type of observations.");
		if (a.has("obs") && a.has("graph-slam")) throw std::runtime_error("Error: argument --obs doesn't apply to relative graph-SLAM.");
		if (a.has("se2") == a.has("se3")) throw std::runtime_error("Exactly one of --se2 or --se3 flags must be set.");
		if ((!a.has("graph-slam") && (a.has("lm-2d") == a.has("lm-3d"))) || (a.has("graph-slam") && (a.has("lm-2d") || a.has("lm-3d")))) throw
			std::runtime_error("Exactly one of --lm-2d or --lm-3d or --graph-slam flags must be set.");
		if (!a.has("dataset")) throw std::runtime_error("Error: --dataset is mandatory.");
		Dataset ds; ds.load(a.str("dataset"));
		const std::string obs = a.str("obs");
		if (a.has("se2") && a.has("graph-slam")) return run<kf2kf_poses::SE2, landmarks::RelativePoses2D, observations::RelativePoses_2D, OPT_GRAPH_SLAM>(a, ds);
		if (a.has("se2") && a.has("lm-2d") && obs == "RangeBearing_2D") return run<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::RangeBearing_2D, RBA_OPTIONS_DEFAULT>(a, ds);
		if (a.has("se2") && a.has("lm-2d") && obs == "Cartesian_2D") return run<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::Cartesian_2D, RBA_OPTIONS_DEFAULT>(a, ds);
		if (a.has("se3") && a.has("lm-3d") && obs == "Cartesian_3D") return run<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::Cartesian_3D, RBA_OPTIONS_DEFAULT>(a, ds);
		if (a.has("se3") && a.has("lm-3d") && obs == "RangeBearing_3D") return run<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::RangeBearing_3D, RBA_OPTIONS_DEFAULT>(a, ds);
		if (a.has("se3") && a.has("lm-3d") && obs == "MonocularCamera") return run<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::MonocularCamera, OPT_CAMERA>(a, ds);
		if (a.has("se3") && a.has("lm-3d") && obs == "StereoCamera") return run<kf2kf_poses::SE3, landmarks::Euclidean3D, observations::StereoCamera, OPT_CAMERA>(a, ds);
		throw std::runtime_error("Sorry: the given combination of pose, point and sensor was not precompiled in this program! (see --list-problems)");
	} catch (std::exception &e) {
		if (std::string(e.what()).size()) std::cerr << e.what() << std::endl;
		return 1;
	}
}

// Exercises the export / configuration side of the header-only front-end without any numeric back-end: key-frames are added with run_local_optimization = false
// and optimize_new_edges_alone = false, so no optimisation is requested. Prints "name value" lines that tests/test_frontend_exports.py checks.
#include <srba.h>
#include <mrpt_lite_apps.h>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
using namespace srba;

struct OPTS : public RBA_OPTIONS_DEFAULT { typedef options::sensor_pose_on_robot_none sensor_pose_on_robot_t; };
typedef RbaEngine<kf2kf_poses::SE2, landmarks::Euclidean2D, observations::RangeBearing_2D, OPTS> rba_t;

static std::string slurp(const std::string &f) { std::ifstream s(f.c_str()); std::stringstream b; b << s.rdbuf(); return b.str(); }

int main(int argc, char **argv) {
	const std::string dir = argc > 1 ? argv[1] : ".";
	rba_t rba; rba.setVerbosityLevel(0);
	rba.parameters.srba.max_tree_depth = 3; rba.parameters.srba.max_optimize_depth = 3; rba.parameters.srba.optimize_new_edges_alone = false;
	rba.parameters.ecp.submap_size = 2; rba.parameters.ecp.min_obs_to_loop_closure = 1;
	// 5 key-frames on a line, each sees landmarks {k, k+1, k+2}; landmark 0 has a known position
	for (int kf = 0; kf < 5; kf++) {
		rba_t::new_kf_observations_t obs;
		for (int l = kf; l < kf + 3; l++) {
			rba_t::new_kf_observation_t o; o.obs.feat_id = l; o.obs.obs_data.range = 1.0 + 0.1 * (l - kf); o.obs.obs_data.yaw = 0.2 * (l - kf);
			if (l == 0) { o.is_fixed = true; o.feat_rel_pos[0] = 1.0; o.feat_rel_pos[1] = 0.0; }
			else if (l - kf == 2 || kf == 0) { o.is_unknown_with_init_val = true; o.feat_rel_pos[0] = 1.0 + 0.1 * (l - kf); o.feat_rel_pos[1] = 0.1; } // first sighting, with an initial value
			obs.push_back(o);
		}
		rba_t::TNewKeyFrameInfo info; rba.define_new_keyframe(obs, info, false /* no optimisation */);
	}
	std::printf("keyframes %zu\nedges %zu\nknown %zu\nunknown %zu\nobservations %zu\n", rba.get_rba_state().keyframes.size(), rba.get_k2k_edges().size(), rba.get_known_feats().size(),
		rba.get_unknown_feats().size(), rba.get_rba_state().all_observations.size());
	// DOT exports
	if (!rba.save_graph_as_dot(dir + "/graph.dot", false) || !rba.save_graph_as_dot(dir + "/graph_lm.dot", true) || !rba.save_graph_top_structure_as_dot(dir + "/top.dot", true)) return 2;
	std::printf("dot_bad_path %d\n", rba.save_graph_as_dot("/nonexistent-dir/x.dot") ? 1 : 0);
	// scene geometry
	rba_t::TOpenGLRepresentationOptions gl; mrpt::opengl::CSetOfObjectsPtr scene = mrpt::opengl::CSetOfObjects::Create(), tree = mrpt::opengl::CSetOfObjects::Create();
	rba.build_opengl_representation(0, gl, scene, tree);
	size_t corners = 0, lines = 0, points = 0, texts = 0;
	for (size_t i = 0; i < scene->objects.size(); i++) {
		if (const mrpt::opengl::CSetOfLines *l = dynamic_cast<const mrpt::opengl::CSetOfLines *>(scene->objects[i].get())) lines += l->size();
		else if (const mrpt::opengl::CPointCloud *p = dynamic_cast<const mrpt::opengl::CPointCloud *>(scene->objects[i].get())) points += p->size();
		else if (dynamic_cast<const mrpt::opengl::CText *>(scene->objects[i].get())) texts++;
		else corners++;
	}
	size_t tree_lines = 0; for (size_t i = 0; i < tree->objects.size(); i++) if (const mrpt::opengl::CSetOfLines *l = dynamic_cast<const mrpt::opengl::CSetOfLines *>(tree->objects[i].get()))
		tree_lines += l->size();
	std::printf("scene_corners %zu\nscene_lines %zu\nscene_points %zu\nscene_texts %zu\ntree_lines %zu\n", corners, lines, points, texts, tree_lines);
	gl.span_tree_max_depth = 1; rba.build_opengl_representation(4, gl, scene);
	size_t corners1 = 0; for (size_t i = 0; i < scene->objects.size(); i++) if (dynamic_cast<const mrpt::opengl::CSetOfObjects *>(scene->objects[i].get())) corners1++;
	std::printf("scene_corners_depth1 %zu\n", corners1);
	// parameter files
	rba.parameters.srba.max_tree_depth = 7; rba.parameters.srba.max_lambda = 1e9; rba.parameters.srba.use_robust_kernel = true; rba.parameters.srba.cov_recovery = crpNone;
		rba.parameters.srba.max_iters = 33;
	rba.parameters.srba.saveToConfigFileName(dir + "/params.ini", "srba"); rba.parameters.ecp.saveToConfigFileName(dir + "/params.ini", "ecp");
	rba_t other; other.parameters.srba.loadFromConfigFileName(dir + "/params.ini", "srba"); other.parameters.ecp.loadFromConfigFileName(dir + "/params.ini", "ecp");
	std::printf("cfg_max_tree_depth %u\ncfg_max_lambda %g\ncfg_robust %d\ncfg_cov %d\ncfg_max_iters %zu\ncfg_submap %zu\ncfg_min_obs %zu\n", (unsigned)other.parameters.srba.max_tree_depth,
		other.parameters.srba.max_lambda, (int)other.parameters.srba.use_robust_kernel,
		(int)other.parameters.srba.cov_recovery, other.parameters.srba.max_iters, other.parameters.ecp.submap_size, other.parameters.ecp.min_obs_to_loop_closure);
	// camera calibration files in the layout of the reference's dataset .cfg
	{ std::ofstream f((dir + "/cam.cfg").c_str());
		f <<
		"[CAMERA_LEFT]\nresolution = [1024 768]\ncx = 512\ncy = 384\nfx = 200\nfy = 150\ndist = [0 0 0 0 0] // K1 K2 T1 T2 K3\n[CAMERA_RIGHT]\nresolution = [1024 768]\ncx=511\n"
			"cy=383\nfx=201\nfy=151\ndist=[0 0 0 0 0]\n[CAMERA_LEFT2RIGHT_POSE]\npose_quaternion = [0.20 0 0  1 0 0 0]   // x y z qr qx qy qz\n"; }
	mrpt::utils::TStereoCamera sc; sc.loadFromConfigFile("CAMERA", mrpt::utils::CConfigFile(dir + "/cam.cfg"));
	std::printf("cam_left %g %g %g %g %u\ncam_right %g %g %g %g\ncam_baseline %g\n", sc.leftCamera.fx(), sc.leftCamera.fy(), sc.leftCamera.cx(), sc.leftCamera.cy(), sc.leftCamera.ncols,
		sc.rightCamera.fx(), sc.rightCamera.fy(), sc.rightCamera.cx(), sc.rightCamera.cy(), sc.rightCameraPose.x());
	// pose algebra used by the tutorials: A - B, quaternion round trip
	const mrpt::poses::CPose3DQuat A(1, 2, 3, mrpt::math::CQuaternionDouble(0.9238795325112867, 0, 0, 0.3826834323650898)), Bq(0.5, -1, 0, mrpt::math::CQuaternionDouble(0.7071067811865476,
		0.7071067811865476, 0, 0));
	const mrpt::poses::CPose3DQuat D = A - Bq; const mrpt::poses::CPose3D back = mrpt::poses::CPose3D(Bq) + mrpt::poses::CPose3D(D);
	std::printf("pose_roundtrip %.3e\n", std::fabs(back.x() - 1) + std::fabs(back.y() - 2) + std::fabs(back.z() - 3) + std::fabs(back.yaw() - M_PI / 4) + std::fabs(back.pitch()) +
		std::fabs(back.roll()));
	Eigen::Matrix<double, 3, 3> L; L.setZero(); L(0, 0) = 4; L(2, 2) = 9; mrpt::math::CMatrixFixed<3, 3> C = L; std::printf("eigen_alias %g %g\n", C(0, 0), C(2, 2));
	mrpt::random::randomGenerator.randomize(7); double s = 0, s2 = 0; for (int i = 0; i < 20000; i++) { const double v = mrpt::random::randomGenerator.drawGaussian1D(2.0, 0.5); s += v; s2 += v * v; }
	std::printf("gauss_mean %.3f\ngauss_std %.3f\n", s / 20000, std::sqrt(s2 / 20000 - (s / 20000) * (s / 20000)));
	return 0;
}

/*
 * srba_types.h -- data model of the host front-end (MI355X build of SRBA's RbaEngine<> API).
 *
 * Mirrors the PUBLIC types of the reference's include/srba/srba_types.h (IDs :21-25, traits :49-156,
 * TNewEdgeInfo :184-200, new_kf_observation_t :473-498, k2f/k2k edges :82-91,:501-525, TSpanTreeEntry :538-542,
 * TRBA_Problem_state :548-785) but stores the problem as index-linked records (edge ids, observation indices)
 * instead of a web of pointers, so that one optimize_edges() call can be flattened into a srba_problem_capsule
 * (include/srba_hip.h) with a few linear passes.  Iteration orders that define unknown numbering and
 * spanning-tree tables (std::map ascending, insertion order of adjacency lists) are preserved.
 */
#pragma once
#include "../mrpt_lite.h"
#include <deque>
#include <map>
#include <set>
#include <vector>

namespace srba {
typedef uint64_t TKeyFrameID;  //!< Numeric IDs for key-frames (KFs)
typedef uint64_t TLandmarkID;  //!< Numeric IDs for landmarks
typedef uint64_t topo_dist_t;  //!< Topological distances
typedef std::pair<TKeyFrameID, TKeyFrameID> TPairKeyFrameID;  //!< directed edge (first --> second)
typedef std::multimap<size_t, TKeyFrameID, std::greater<size_t> > base_sorted_lst_t;  //!< KFs sorted in descending order by some count

#define SRBA_INVALID_KEYFRAMEID static_cast<srba::TKeyFrameID>(-1)
#define SRBA_INVALID_INDEX static_cast<size_t>(-1)

template <class landmark_t, class obs_t> struct sensor_model;

/** Covariance recovery policy (reference srba_types.h:164-169) */
enum TCovarianceRecoveryPolicy { crpNone = 0, crpLandmarksApprox };

template <typename PAIR, typename V> V getTheOtherFromPair(const V one, const PAIR &p) { return p.first == one ? p.second : p.first; }
template <typename K2K_EDGE, typename V> V getTheOtherFromPair2(const V one, const K2K_EDGE &p) { return p.from == one ? p.to : p.from; }

/** Used in TNewKeyFrameInfo (reference srba_types.h:184-200) */
struct TNewEdgeInfo {
	size_t id;                 //!< The new edge ID
	bool has_approx_init_val;  //!< Whether the edge was assigned an approximated initial value
	TKeyFrameID loopclosure_observer_kf, loopclosure_base_kf; //!< loop-closure helpers filled by the ECP
	TNewEdgeInfo() : id(SRBA_INVALID_INDEX), has_approx_init_val(false), loopclosure_observer_kf(SRBA_INVALID_KEYFRAMEID), loopclosure_base_kf(SRBA_INVALID_KEYFRAMEID) {}
};

template <class POSE_TRAITS> struct kf2kf_pose_traits : public POSE_TRAITS {
	typedef typename POSE_TRAITS::pose_t pose_t;
	typedef mrpt::math::CArrayDouble<POSE_TRAITS::REL_POSE_DIMS> array_pose_t;
	/** relative pose + "up-to-date" flag (reference :57-68) */
	struct pose_flag_t { pose_t pose; mutable bool updated; pose_flag_t() : updated(false) {} void mark_outdated() const { updated = false; } };
	typedef std::map<TKeyFrameID, pose_flag_t> frameid2pose_map_t;
	/** Keyframe-to-keyframe edge: an unknown of the problem (reference :82-91) */
	struct k2k_edge_t { TKeyFrameID from, to; pose_t inv_pose; /*!< pose of "from" as seen from "to" */ size_t id; };
};

template <class LM_TRAITS> struct landmark_traits : public LM_TRAITS {
	typedef mrpt::math::CArrayDouble<LM_TRAITS::LM_DIMS> array_landmark_t;
	struct TRelativeLandmarkPos {
		TRelativeLandmarkPos() : id_frame_base(SRBA_INVALID_KEYFRAMEID) {}
		template <typename LANDMARK_POS> TRelativeLandmarkPos(const TKeyFrameID b, const LANDMARK_POS &p) : id_frame_base(b) { for (size_t i = 0; i < LM_TRAITS::LM_DIMS; i++) pos[i] = p[i]; }
		TKeyFrameID id_frame_base;  //!< base KF of the landmark
		array_landmark_t pos;       //!< parameters relative to the base KF
	};
	typedef std::map<TLandmarkID, TRelativeLandmarkPos> TRelativeLandmarkPosMap;
	struct TLandmarkEntry { bool has_known_pos; TRelativeLandmarkPos *rfp; TLandmarkEntry() : has_known_pos(true), rfp(NULL) {} TLandmarkEntry(bool k, TRelativeLandmarkPos *r) : has_known_pos(k), rfp(r) {} };
};

template <class OBS_TRAITS> struct observation_traits : public OBS_TRAITS {
	typedef mrpt::math::CArrayDouble<OBS_TRAITS::OBS_DIMS> array_obs_t;
	typedef mrpt::math::CArrayDouble<OBS_TRAITS::OBS_DIMS> residual_t;
	typedef std::vector<residual_t> vector_residuals_t;
	struct observation_t { TLandmarkID feat_id; typename OBS_TRAITS::obs_data_t obs_data; observation_t() : feat_id(0), obs_data() {} };
};

/** Types depending on the <pose,landmark,observation> triple (reference :437-534) */
template <class kf2kf_pose_t, class landmark_t, class obs_t> struct rba_joint_parameterization_traits_t {
	typedef landmark_t original_landmark_t; typedef kf2kf_pose_t original_kf2kf_pose_t;
	typedef kf2kf_pose_traits<kf2kf_pose_t> kf2kf_traits_t; typedef observation_traits<obs_t> obs_traits_t; typedef landmark_traits<landmark_t> lm_traits_t;
	typedef typename kf2kf_traits_t::k2k_edge_t k2k_edge_t;
	/** One observation from a new KF as given by the user (reference :473-495) */
	struct new_kf_observation_t {
		new_kf_observation_t() : is_fixed(false), is_unknown_with_init_val(false) { feat_rel_pos.setZero(); }
		typename obs_traits_t::observation_t obs;
		bool is_fixed;                  //!< first observation of a landmark with fixed (known) relative position
		bool is_unknown_with_init_val;  //!< first observation of an unknown landmark whose initial value is in feat_rel_pos
		typename lm_traits_t::array_landmark_t feat_rel_pos;
		template <class REL_POS> void setRelPos(const REL_POS &pos) { for (size_t i = 0; i < landmark_t::LM_DIMS; i++) feat_rel_pos[i] = pos[i]; }
	};
	typedef std::deque<new_kf_observation_t> new_kf_observations_t;
	struct kf_observation_t { typename obs_traits_t::observation_t obs; typename obs_traits_t::array_obs_t obs_arr; TKeyFrameID kf_id; };
	/** Keyframe-to-feature edge (reference :515-525) */
	struct k2f_edge_t {
		kf_observation_t obs; bool feat_has_known_rel_pos; bool is_first_obs_of_unknown;
		typename lm_traits_t::TRelativeLandmarkPos *feat_rel_pos;
		TLandmarkID get_observed_feature_id() const { return obs.obs.feat_id; }
	};
	struct keyframe_info { std::deque<k2k_edge_t *> adjacent_k2k_edges; std::deque<k2f_edge_t *> adjacent_k2f_edges; };
};

/** Entry of the symbolic spanning trees (reference :538-542) */
struct TSpanTreeEntry { TKeyFrameID next; topo_dist_t distance; };

/** Symbolic record of one dh_dAp block (reference TJacobianSymbolicInfo_dh_dAp :204-240), index-linked. */
struct TJacobianSymbolicInfo_dh_dAp {
	size_t obs_idx;        //!< global observation index (row)
	size_t k2k_edge_id;    //!< column
	TKeyFrameID kf_d;      //!< "d+1": node on the observer side of the edge
	TKeyFrameID kf_base;   //!< base KF of the observed landmark
	bool edge_normal_dir;  //!< edge.to == kf_d
	bool has_A;            //!< false when kf_d is the observer itself (rel_pose_d1_from_obs == NULL)
};
/** Symbolic record of one dh_df block (reference :245-266) */
struct TJacobianSymbolicInfo_dh_df { size_t obs_idx; bool has_pose; /*!< false for the first observation (observer == base) */ };

} // namespace srba

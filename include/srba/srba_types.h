/*
 * srba_types.h -- public value types of the srba:: front-end (MI355X build).
 *
 * The NAMES and FIELDS a user program touches are those of the reference (include/srba/srba_types.h: identifiers :21-25, TNewEdgeInfo :184-200,
 * new_kf_observation_t :473-498, k2k_edge_t :82-91, kf_observation_t / k2f_edge_t :501-525, TRelativeLandmarkPos :106-121), so that code
 * written against MRPT/srba compiles unchanged.  What differs is the storage behind them: the reference links these records with raw
 * pointers inside std::map / std::deque nodes; here they are plain records in flat arrays, linked by 32-bit indices that live in
 * srba::graph::topology (graph_topology.h).  Pointer-valued fields of the reference that user code may read (k2f_edge_t::feat_rel_pos)
 * are materialised on access by the views in RbaEngine.h.
 */
#pragma once
#include "../mrpt_lite.h"
#include <cstdint>
#include <deque>
#include <map>
#include <utility>
#include <vector>

namespace srba {

typedef uint64_t TKeyFrameID;
typedef uint64_t TLandmarkID;
typedef uint64_t topo_dist_t;
typedef std::pair<TKeyFrameID, TKeyFrameID> TPairKeyFrameID; // (from, to)
/** (count, key-frame) list ordered by decreasing count; equal counts keep insertion order */
typedef std::multimap<size_t, TKeyFrameID, std::greater<size_t> > base_sorted_lst_t;

#define SRBA_INVALID_KEYFRAMEID static_cast<srba::TKeyFrameID>(-1)
#define SRBA_INVALID_INDEX static_cast<size_t>(-1)

template <class landmark_t, class obs_t> struct sensor_model;

enum TCovarianceRecoveryPolicy { crpNone = 0, crpLandmarksApprox };

template <typename PAIR, typename V> V getTheOtherFromPair(const V one, const PAIR &p) { return p.first == one ? p.second : p.first; }
template <typename K2K_EDGE, typename V> V getTheOtherFromPair2(const V one, const K2K_EDGE &p) { return p.from == one ? p.to : p.from; }

/** What define_new_keyframe() reports about each kf2kf edge it created. */
struct TNewEdgeInfo {
	size_t id;
	bool has_approx_init_val;                                  // an initial relative pose could be derived
	TKeyFrameID loopclosure_observer_kf, loopclosure_base_kf;  // set by the edge-creation policy for loop-closure edges between two area centres
	TNewEdgeInfo() : id(SRBA_INVALID_INDEX), has_approx_init_val(false), loopclosure_observer_kf(SRBA_INVALID_KEYFRAMEID), loopclosure_base_kf(SRBA_INVALID_KEYFRAMEID) {}
};

template <class POSE_TRAITS> struct kf2kf_pose_traits : public POSE_TRAITS {
	typedef typename POSE_TRAITS::pose_t pose_t;
	typedef mrpt::math::CArrayDouble<POSE_TRAITS::REL_POSE_DIMS> array_pose_t;
	/** a numeric spanning-tree pose and whether it reflects the current edge values */
	struct pose_flag_t { pose_t pose; mutable bool updated; pose_flag_t() : updated(false) {} void mark_outdated() const { updated = false; } };
	typedef std::map<TKeyFrameID, pose_flag_t> frameid2pose_map_t; // result type of create_complete_spanning_tree()
	/** kf2kf edge = one unknown relative pose; inv_pose is the pose of `from` as seen from `to`; id = position in get_k2k_edges() */
	struct k2k_edge_t { TKeyFrameID from, to; pose_t inv_pose; size_t id; };
	typedef std::vector<k2k_edge_t> k2k_edge_vector_t;
};

template <class LM_TRAITS> struct landmark_traits : public LM_TRAITS {
	typedef mrpt::math::CArrayDouble<LM_TRAITS::LM_DIMS> array_landmark_t;
	/** landmark coordinates relative to its base key-frame */
	struct TRelativeLandmarkPos {
		TKeyFrameID id_frame_base; array_landmark_t pos;
		TRelativeLandmarkPos() : id_frame_base(SRBA_INVALID_KEYFRAMEID) { pos.setZero(); }
		template <typename LANDMARK_POS> TRelativeLandmarkPos(const TKeyFrameID b, const LANDMARK_POS &p) : id_frame_base(b) { for (size_t i = 0; i < LM_TRAITS::LM_DIMS; i++) pos[i] = p[i]; }
	};
};

template <class OBS_TRAITS> struct observation_traits : public OBS_TRAITS {
	typedef mrpt::math::CArrayDouble<OBS_TRAITS::OBS_DIMS> array_obs_t;
	typedef mrpt::math::CArrayDouble<OBS_TRAITS::OBS_DIMS> residual_t;
	typedef std::vector<residual_t> vector_residuals_t;
	struct observation_t { TLandmarkID feat_id; typename OBS_TRAITS::obs_data_t obs_data; observation_t() : feat_id(0), obs_data() {} };
};

template <class kf2kf_pose_t, class landmark_t, class obs_t> struct rba_joint_parameterization_traits_t {
	typedef landmark_t original_landmark_t; typedef kf2kf_pose_t original_kf2kf_pose_t;
	typedef kf2kf_pose_traits<kf2kf_pose_t> kf2kf_traits_t; typedef observation_traits<obs_t> obs_traits_t; typedef landmark_traits<landmark_t> lm_traits_t;
	typedef typename kf2kf_traits_t::k2k_edge_t k2k_edge_t;
	/** one entry of the list handed to define_new_keyframe() */
	struct new_kf_observation_t {
		typename obs_traits_t::observation_t obs;
		bool is_fixed;                 // first sighting of a landmark whose position relative to this key-frame is known: feat_rel_pos
		bool is_unknown_with_init_val; // first sighting of an unknown landmark with a caller-supplied initial position: feat_rel_pos
		typename lm_traits_t::array_landmark_t feat_rel_pos;
		new_kf_observation_t() : is_fixed(false), is_unknown_with_init_val(false) { feat_rel_pos.setZero(); }
		template <class REL_POS> void setRelPos(const REL_POS &pos) { for (size_t i = 0; i < landmark_t::LM_DIMS; i++) feat_rel_pos[i] = pos[i]; }
	};
	typedef std::deque<new_kf_observation_t> new_kf_observations_t;
	/** stored observation: the user's record, its numeric vector and the observing key-frame */
	struct kf_observation_t { typename obs_traits_t::observation_t obs; typename obs_traits_t::array_obs_t obs_arr; TKeyFrameID kf_id; };
	/** key-frame -> feature edge as user code sees it (built on access from the flat tables) */
	struct k2f_edge_t {
		kf_observation_t obs; bool feat_has_known_rel_pos, is_first_obs_of_unknown;
		const typename lm_traits_t::TRelativeLandmarkPos *feat_rel_pos;
		TLandmarkID get_observed_feature_id() const { return obs.obs.feat_id; }
	};
};

/** (next hop, distance) of one symbolic spanning-tree entry, as returned by TSpanningTree look-ups */
struct TSpanTreeEntry { TKeyFrameID next; topo_dist_t distance; };

} // namespace srba

/*
 * srba.h -- umbrella include, same name as the reference's include/srba.h:15-23.
 *   #include <srba.h>   then   srba::RbaEngine<kf2kf_poses::SE2, landmarks::RelativePoses2D, observations::RelativePoses_2D, OPTS>
 * The numeric optimiser behind define_new_keyframe()/optimize_local_area() runs on an MI355X (see srba_hip.h).
 */
#pragma once
#include "mrpt_lite.h"
#include "srba/srba_types.h"
#include "srba/models.h"
#include "srba/srba_options.h"
#include "srba/capsule.h"
#include "srba/RbaEngine.h"
#include "srba/hip_backend.h"

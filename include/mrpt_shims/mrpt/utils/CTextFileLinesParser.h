#pragma once
#include <mrpt_lite_apps.h>

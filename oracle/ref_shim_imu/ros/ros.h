// Build shim (OURS): the ROS logging macros src/IMU_Processing.hpp mentions - silent here (and the standard headers the real ROS / PCL headers pull in for it: <algorithm>, <iomanip>, streams).  Test infrastructure only.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_ASSERT(x) assert(x)

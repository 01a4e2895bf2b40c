// Build shim (OURS): src/IMU_Processing.hpp includes this header and uses nothing of it.
#pragma once

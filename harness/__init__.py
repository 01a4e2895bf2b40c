"""Test / benchmark harness (NOT part of the product path): synthetic worlds, sensors and driver messages, and ROS-free
numpy stand-ins for the host code the reference keeps around the hot path (constant-velocity and IMU forward propagation).
Used by tests/, bench.py and __graft_entry__.smoke() only; the library in lidar_imu_init_amd/ never imports it."""

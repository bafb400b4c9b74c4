// internal: device-wide int32 exclusive scan (scan.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
size_t ms3d_scan_workspace_bytes();
// out[i] = sum(in[0..i)), in == out allowed; *total_out_dev (device, optional) = sum of all
int ms3d_exclusive_scan_i32(const int *in, int *out, int n, int *total_out_dev, void *workspace, hipStream_t stream);

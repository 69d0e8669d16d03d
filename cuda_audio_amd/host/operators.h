// operators.h — the small part of the reference's operators.h that its hot
// path used (dim3 arithmetic for the grid-stride idiom,
// float2 +, -, * scalar, / scalar, clamp), written from scratch for host code.
// The reference's file is NVIDIA's helper_math.h under NVIDIA's EULA and is
// deliberately not reproduced; the HIP kernels of this engine do not need it
// (HIP's float2 already has component-wise operators).
#pragma once
#include <algorithm>

#include "wav.h"

struct host_dim3 {
    unsigned x = 1, y = 1, z = 1;
};
struct host_uint3 {  // what threadIdx / blockIdx are in the reference's kernels
    unsigned x = 0, y = 0, z = 0;
};
// the grid-stride idiom of the reference's kernels: offset = blockDim * blockIdx + threadIdx, stride = blockDim * gridDim
inline host_dim3 operator*(host_dim3 a, host_dim3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }   // operators.h:74
inline host_dim3 operator*(host_dim3 a, host_uint3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // operators.h:84
inline host_dim3 operator+(host_dim3 a, host_uint3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }  // operators.h:99

inline wav_float2 operator+(wav_float2 a, wav_float2 b) { return {a.x + b.x, a.y + b.y}; }
inline wav_float2 operator-(wav_float2 a, wav_float2 b) { return {a.x - b.x, a.y - b.y}; }
inline wav_float2 operator*(wav_float2 a, float s) { return {a.x * s, a.y * s}; }
inline wav_float2 operator*(float s, wav_float2 a) { return {a.x * s, a.y * s}; }
inline wav_float2 operator/(wav_float2 a, float s) { return {a.x / s, a.y / s}; }
inline wav_float2& operator+=(wav_float2& a, wav_float2 b) {
    a.x += b.x;
    a.y += b.y;
    return a;
}
inline wav_float2& operator+=(wav_float2& a, float s) {  // scalar added to both components
    a.x += s;
    a.y += s;
    return a;
}
inline wav_float2 clamp(wav_float2 v, float lo, float hi) {
    return {std::min(std::max(v.x, lo), hi), std::min(std::max(v.y, lo), hi)};
}

// gpu.h — device selection (reference src/gpu.h:17 selectGpu()).  The
// reference scores CUDA devices by SMs x cores/SM x clock from an NVIDIA table;
// here the score is CUs x clock from hipDeviceProp_t, and MCCONV_DEVICE
// overrides the choice.  Returns the selected HIP device ordinal.
#pragma once
int selectGpu();

// placeholder
#ifndef K_BUILD_H_
#define K_BUILD_H_
#include "device_common.h"
DEV void build_round(const JobParams& J, const ShardDesc& D, ShardState* S,
                     const DeviceTables* T, const uint8_t* input, uint8_t* ws) {}
#endif

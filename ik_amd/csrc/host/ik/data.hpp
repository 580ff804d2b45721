// Forwarding header: the reference's include "ik/data.hpp" resolves to the GPU-backed mirror.
#pragma once
#include "ik/ik_gpu.hpp"

// Forwarding header: the reference's include "ik/constraint.hpp" resolves to the GPU-backed mirror.
#pragma once
#include "ik/ik_gpu.hpp"

// Forwarding header: the reference's include "ik/centre_of_mass.hpp" resolves to the GPU-backed mirror.
#pragma once
#include "ik/ik_gpu.hpp"

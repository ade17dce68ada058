#include "../cub.cuh"

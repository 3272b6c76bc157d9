// Forced-included before the reference headers: the reference's image buffers come from malloc (imageNd.hpp:177) and
// several algorithms read parts of them that were never written.  The canonical value of such bytes is 0.
#pragma once
#include <cstdlib>
#include <stdlib.h>
#define malloc(x) calloc(1, (x))

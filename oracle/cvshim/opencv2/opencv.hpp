// cvshim stub: see ../cvshim.hpp (test infrastructure only)
#include "../cvshim.hpp"

#include "../cvshim.hpp"

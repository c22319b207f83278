#include "../../cvshim.hpp"

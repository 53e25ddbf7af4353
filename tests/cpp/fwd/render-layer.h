// forwarding header: the reference's file name -> this repository's drop-in (see include/pbrlab_hip.hpp)
#include "pbrlab_hip.hpp"

// Forwarder to the single PCL/Boost/OpenNI stand-in header used ONLY to build the reference
// Integrate sources as a CPU oracle (test infrastructure; see oracle/README.md).
#pragma once
#include "er_oracle_stub.h"

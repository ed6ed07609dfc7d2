// Forwarder to the PCL stand-in used ONLY to build the reference's BuildCorrespondence / RansacCurvature sources in place as
// a CPU oracle (test infrastructure; see oracle/README.md and er_corres_stub.h).
#pragma once
#include "er_corres_stub.h"

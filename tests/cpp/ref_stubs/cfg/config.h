// stand-in for the config.h the reference's g2o build generates with cmake (Thirdparty/g2o/config.h.in)
#pragma once
#define G2O_OPENMP 0

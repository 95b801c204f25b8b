#pragma once
#include <boost/archive/text_oarchive.hpp>

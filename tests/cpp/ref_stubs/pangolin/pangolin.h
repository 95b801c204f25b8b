// COMPILE-ONLY stand-in: the one Pangolin type the reference's MapDrawer.h mentions.
#pragma once
namespace pangolin { struct OpenGlMatrix { double m[16]; }; }

// Forced include of tests/test_glue_typecheck.py.  The reference's WebSocket.h defines its Boost.Beast / Asio classes
// inline (code/include/WebSocket.h:35-470) - the one header of the tree that cannot be satisfied by declaring a few
// third-party names - so the type-check skips it through its include guard (-DEDGE_SLAM_WEBSOCKET_H) and declares here
// the four names the service headers take from it.  Everything the glue touches (Frame, KeyFrame, MapPoint, Map,
// ORBmatcher, Optimizer and what they include) is the reference's own text.
#pragma once
#include <memory>
#include <string>
#include <sys/types.h>
namespace ORB_SLAM2 {
struct Request { id_t src; id_t dst; std::string path; std::string body; std::string toString() const; };
class ConnectionService {};
namespace WS { namespace Server { class listener; } namespace Client { class session; } }
}

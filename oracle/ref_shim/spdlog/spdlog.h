// oracle/_ref shim: the reference includes spdlog but the ORB extractor path logs nothing
#pragma once
namespace spdlog { template <typename... A> inline void info(A&&...) {} template <typename... A> inline void debug(A&&...) {} template <typename... A> inline void warn(A&&...) {} }

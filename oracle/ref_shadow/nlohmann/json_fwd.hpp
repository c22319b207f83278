#pragma once
namespace nlohmann { class json; }

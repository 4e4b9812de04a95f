// placeholder replaced below
#include "mdb_common.hpp"
using namespace mdb;
extern "C" {
int mdb_compress_series(mdb_ctx *, const int64_t *, const float *, uint64_t, mdb_error_bound, mdb_segments_owned **) { return fail("not built yet"); }
int mdb_compress_chunks(mdb_ctx *, const int64_t *, const float *, const uint64_t *, uint64_t, mdb_error_bound, mdb_segments_owned **) { return fail("not built yet"); }
int mdb_compress_chunks_dev(mdb_ctx *, const int64_t *, const float *, const uint64_t *, uint64_t, mdb_error_bound, int64_t, int64_t, const uint64_t *, mdb_segments_owned **) { return fail("not built yet"); }
}

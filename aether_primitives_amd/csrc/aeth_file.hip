// aeth_file.hip -- raw sample files (reference: src/util/file.rs:12-107) and the
// file -> device -> file FIR pipeline.  SURVEY 8f "next" rows #3 and #4.
//
// The reference's binary format is a header-less, native-endian dump of back-to-back
// structs (BinaryWriter::write casts the slice to bytes, util/file.rs:101-109): for cf32
// that is byte for byte the layout of a device buffer, so a recorded IQ stream can be
// mapped and pushed through the GPU without any conversion.
#include "aeth_internal.h"
#include "aeth_fft_plan.h"

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" {

/* count_structs_in_file (util/file.rs:12-25) */
int aeth_file_count_structs(const char *path, size_t elem_size, size_t *count)
{
    AETH_REQUIRE(path && count && elem_size, AETH_E_ARG, "null/zero argument");
    struct stat st;
    AETH_REQUIRE(stat(path, &st) == 0, AETH_E_ARG, "%s: %s", path, strerror(errno));
    AETH_REQUIRE((size_t)st.st_size % elem_size == 0, AETH_E_LEN,
                 "File does not contain an integer number of the requested struct");   /* :20-23 */
    *count = (size_t)st.st_size / elem_size;
    return AETH_OK;
}

/* BinaryReader::read (util/file.rs:46-57): fill `n` structs starting at struct `offset` */
int aeth_file_read(const char *path, size_t offset, void *dst, size_t n, size_t elem_size)
{
    AETH_REQUIRE(path && (dst || !n) && elem_size, AETH_E_ARG, "null/zero argument");
    FILE *f = fopen(path, "rb");
    AETH_REQUIRE(f, AETH_E_ARG, "%s: %s", path, strerror(errno));
    int rc = AETH_OK;
    if (fseeko(f, (off_t)(offset * elem_size), SEEK_SET) != 0 || fread(dst, elem_size, n, f) != n)
        rc = aeth::set_error(AETH_E_LEN, "failed to fill whole buffer");               /* read_exact's UnexpectedEof */
    fclose(f);
    return rc;
}

/* binary_writer + BinaryWriter::write (util/file.rs:83-109): create/truncate unless append */
int aeth_file_write(const char *path, const void *src, size_t n, size_t elem_size, int append)
{
    AETH_REQUIRE(path && (src || !n) && elem_size, AETH_E_ARG, "null/zero argument");
    FILE *f = fopen(path, append ? "ab" : "wb");
    AETH_REQUIRE(f, AETH_E_ARG, "%s: %s", path, strerror(errno));
    int rc = AETH_OK;
    if (fwrite(src, elem_size, n, f) != n) rc = aeth::set_error(AETH_E_ARG, "%s: short write", path);
    if (fclose(f) != 0 && rc == AETH_OK) rc = aeth::set_error(AETH_E_ARG, "%s: %s", path, strerror(errno));
    return rc;
}

/* raw cf32 file -> one of the pipeline's ops -> raw file (cf32 samples, or bit bytes for the demodulating stage): both
 * files are mapped and handed to the host-stream pipeline (aeth_stream_host) */
int aeth_stream_file(aeth_ctx *ctx, const aeth_stream_op *op, const char *in_path, const char *out_path, size_t chunk,
                     aeth_pipe_stats *stats)
{
    AETH_REQUIRE(ctx && op && in_path && out_path, AETH_E_ARG, "null argument");
    if (stats) *stats = aeth_pipe_stats{0, 0, 0, 0};
    // the input is opened, measured and mapped BEFORE the output is created or truncated, and a path that names
    // the same file twice is refused: truncating the recording before reading it would process zeros
    int fi = open(in_path, O_RDONLY);
    AETH_REQUIRE(fi >= 0, AETH_E_ARG, "%s: %s", in_path, strerror(errno));
    struct stat si;
    if (fstat(fi, &si) != 0) { int e = errno; close(fi); return aeth::set_error(AETH_E_ARG, "%s: %s", in_path, strerror(e)); }
    struct stat so;
    if (stat(out_path, &so) == 0 && so.st_dev == si.st_dev && so.st_ino == si.st_ino) {
        close(fi);
        return aeth::set_error(AETH_E_ARG, "%s and %s are the same file: the stream cannot run in place", in_path, out_path);
    }
    if ((size_t)si.st_size % sizeof(aeth_cf32) != 0) {
        close(fi);
        return aeth::set_error(AETH_E_LEN, "%s: %lld bytes is not a whole number of cf32 samples", in_path, (long long)si.st_size);
    }
    const size_t n = (size_t)si.st_size / sizeof(aeth_cf32);
    const size_t bytes = n * sizeof(aeth_cf32);
    const size_t n_out = aeth_stream_out_count(ctx, op, n);
    const size_t out_unit = op->kind == AETH_STREAM_FFT_MUL_IFFT_DEMOD ? 1 : sizeof(aeth_cf32);
    const size_t obytes = n_out * out_unit;
    if (n > 0 && n_out == 0) { close(fi); return aeth::set_error(AETH_E_ARG, "bad stream op"); }
    void *mi = MAP_FAILED, *mo = MAP_FAILED;
    int rc = AETH_OK;
    if (n > 0) {
        mi = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fi, 0);
        if (mi == MAP_FAILED) { int e = errno; close(fi); return aeth::set_error(AETH_E_NOMEM, "mmap %s: %s", in_path, strerror(e)); }
    }
    int fo = open(out_path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fo < 0) rc = aeth::set_error(AETH_E_ARG, "%s: %s", out_path, strerror(errno));
    if (rc == AETH_OK && n > 0) {
        if (ftruncate(fo, (off_t)obytes) != 0) rc = aeth::set_error(AETH_E_ARG, "%s: %s", out_path, strerror(errno));
        if (rc == AETH_OK) {
            mo = mmap(nullptr, obytes, PROT_READ | PROT_WRITE, MAP_SHARED, fo, 0);
            if (mo == MAP_FAILED) rc = aeth::set_error(AETH_E_NOMEM, "mmap %s: %s", out_path, strerror(errno));
        }
        if (rc == AETH_OK) rc = aeth_stream_host(ctx, op, mi, n, mo, n_out, chunk, stats);
    }
    if (mi != MAP_FAILED) munmap(mi, bytes);
    if (mo != MAP_FAILED) { msync(mo, obytes, MS_SYNC); munmap(mo, obytes); }
    close(fi);
    if (fo >= 0) close(fo);
    return rc;
}

/* raw cf32 file -> FIR -> raw cf32 file */
int aeth_fir_stream_file(aeth_fir *fir, const char *in_path, const char *out_path, size_t chunk, aeth_pipe_stats *stats)
{
    AETH_REQUIRE(fir && in_path && out_path, AETH_E_ARG, "null argument");
    aeth_stream_op op{};
    op.kind = AETH_STREAM_FIR; op.fir = fir;
    return aeth_stream_file(fir->ctx, &op, in_path, out_path, chunk, stats);
}

}  // extern "C"

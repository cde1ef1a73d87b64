// jpeg-encoder: command-line front end with the reference's interface
// (src/main.rs:8-68, src/utils/arguments.rs:6-66).  Same flags, aliases,
// defaults, printed lines and exit codes; the five JpegImage calls of main()
// become the five staged calls of libjpeg_hip.so.  Extra flags (not in the
// reference): --device N, --fused, --stripes N with --devices a,b,.. (one image cut
// into MCU-row stripes coded on several GPUs), and --image given more than once (a batch).
#include <cstdint>
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "jpeg_hip.h"
#include "jpeg_host.h"

namespace {

constexpr int EXIT_PANIC = 101;   // a Rust panic in the reference
constexpr int EXIT_USAGE = 2;     // a clap parse error in the reference

const char *USAGE =
    "Usage: jpeg-encoder [OPTIONS] --image <IMAGE>\n"
    "\n"
    "Options:\n"
    "  -i, --image <IMAGE>                          the input image path\n"
    "  -o, --output <OUTPUT>                        the output image path (optional) [default: ]\n"
    "  -s, --subsampling-ratio <SUBSAMPLING_RATIO>  Subsampling ratio in the format `4:2:0`, `4:4:4`, or `4:2:2` "
    "[default: 4:2:0]\n"
    "  -d, --dct-algorithm <DCT_ALGORITHM>          DCT algorithm to use: either \"RealDct\" or \"BinDct\" "
    "[default: real-dct] [possible values: real-dct, bin-dct]\n"
    "                                               (real-dct sums each coefficient in two 8-point passes: within +-1 of\n"
    "                                               the reference in at most 0.01 % of coefficients; real-dct-strict, not a\n"
    "                                               value of the reference, keeps its summation order and is bit-exact)\n"
    "      --device <N>                             GPU to use [default: 0]\n"
    "      --fused                                  one fused device pass instead of the five stages\n"
    "      --stripes <N>                            cut the image into N MCU-row stripes coded independently\n"
    "      --devices <A,B,..>                       GPUs the stripes (or the images of a batch) are spread over\n"
    "                                               (--image may be given several times: a batch)\n"
    "  -h, --help                                   Print help\n"
    "  -V, --version                                Print version\n";

[[noreturn]] void usage_error(const std::string &msg) {
    std::fprintf(stderr, "error: %s\n\n%s\nFor more information, try '--help'.\n", msg.c_str(), USAGE);
    std::exit(EXIT_USAGE);
}

[[noreturn]] void panic(const char *msg) {
    std::fprintf(stderr, "thread 'main' panicked:\n%s\n", msg);
    std::exit(EXIT_PANIC);
}

struct Args {
    std::string output, ratio_text = "4:2:0", dct = "real-dct";
    std::vector<std::string> images;   // all --image values
    int ratio[3] = {4, 2, 0};
    int device = 0;
    std::vector<int> devices;
    int stripes = 0;
    bool fused = false;
};

// parse_subsampling_ratio (src/utils/arguments.rs:25-43)
void parse_ratio(const std::string &s, int out[3]) {
    std::vector<std::string> parts;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(':', a);
        parts.push_back(s.substr(a, b == std::string::npos ? b : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    if (parts.size() != 3)
        usage_error("invalid value '" + s + "' for '--subsampling-ratio <SUBSAMPLING_RATIO>': Subsampling ratio must be "
                    "in the format A:B:C");
    for (int k = 0; k < 3; k++) {
        const std::string &p = parts[k];
        // Rust's u8::from_str: an optional '+', then digits, value <= 255
        size_t i = (!p.empty() && p[0] == '+') ? 1 : 0;
        bool ok = i < p.size();
        long v = 0;
        for (; ok && i < p.size(); i++) {
            if (p[i] < '0' || p[i] > '9') ok = false;
            else v = v * 10 + (p[i] - '0');
            if (v > 255) ok = false;
        }
        if (!ok)
            usage_error("invalid value '" + s + "' for '--subsampling-ratio <SUBSAMPLING_RATIO>': Subsampling ratio "
                        "must consist of three integers separated by colons");
        out[k] = (int)v;
    }
    // get_downsampling_factor panics inside the parser (src/utils/arguments.rs:37, src/jpeg/sampling.rs:6-22)
    const bool valid = out[0] == 4 && ((out[1] == 4 && out[2] == 4) || (out[1] == 2 && (out[2] == 0 || out[2] == 2)));
    if (!valid) panic(jpg_strerror(JPG_ERR_INVALID_RATIO));
}

Args parse(int argc, char **argv) {
    Args a;
    bool have_image = false;
    auto value = [&](int &i, const char *name, const char *inline_val) -> std::string {
        if (inline_val) return inline_val;
        if (i + 1 >= argc) usage_error(std::string("a value is required for '") + name + "' but none was supplied");
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        std::string key = arg;
        const char *inl = nullptr;
        if (arg.rfind("--", 0) == 0) {
            const size_t eq = arg.find('=');
            if (eq != std::string::npos) {
                key = arg.substr(0, eq);
                inl = argv[i] + eq + 1;
            }
        } else if (arg.size() > 2 && arg[0] == '-') {   // -ifile.bmp or -i=file.bmp
            key = arg.substr(0, 2);
            inl = argv[i] + 2;
            if (*inl == '=') inl++;
        }
        if (key == "-h" || key == "--help") {
            std::printf("%s", USAGE);
            std::exit(0);
        } else if (key == "-V" || key == "--version") {
            std::printf("jpeg-encoder 0.1.0\n");
            std::exit(0);
        } else if (key == "-i" || key == "--image") {
            a.images.push_back(value(i, "--image <IMAGE>", inl));
            have_image = true;
        } else if (key == "-o" || key == "--output") {
            a.output = value(i, "--output <OUTPUT>", inl);
        } else if (key == "-s" || key == "--subsampling-ratio") {
            a.ratio_text = value(i, "--subsampling-ratio <SUBSAMPLING_RATIO>", inl);
        } else if (key == "-d" || key == "--dct-algorithm") {
            a.dct = value(i, "--dct-algorithm <DCT_ALGORITHM>", inl);
            if (a.dct != "real-dct" && a.dct != "bin-dct" && a.dct != "real-dct-strict")
                usage_error("invalid value '" + a.dct + "' for '--dct-algorithm <DCT_ALGORITHM>'\n  [possible values: "
                            "real-dct, bin-dct, real-dct-strict]");
        } else if (key == "--device") {
            a.device = std::atoi(value(i, "--device <N>", inl).c_str());
        } else if (key == "--fused") {
            a.fused = true;
        } else if (key == "--stripes") {
            a.stripes = std::atoi(value(i, "--stripes <N>", inl).c_str());
            if (a.stripes < 1 || a.stripes > 4096) usage_error("invalid value for '--stripes <N>'");
        } else if (key == "--devices") {
            const std::string list = value(i, "--devices <A,B,..>", inl);
            size_t p0 = 0;
            while (p0 <= list.size()) {
                const size_t p1 = list.find(',', p0);
                const std::string tok = list.substr(p0, p1 == std::string::npos ? p1 : p1 - p0);
                if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos)
                    usage_error("invalid value '" + list + "' for '--devices <A,B,..>'");
                a.devices.push_back(std::atoi(tok.c_str()));
                if (p1 == std::string::npos) break;
                p0 = p1 + 1;
            }
        } else {
            usage_error("unexpected argument '" + arg + "' found");
        }
    }
    if (!have_image) usage_error("the following required arguments were not provided:\n  --image <IMAGE>");
    parse_ratio(a.ratio_text, a.ratio);
    if (a.devices.empty()) a.devices.push_back(a.device);
    if (a.images.size() > 1 && !a.output.empty()) usage_error("'--output <OUTPUT>' cannot be used with several images");
    // Args::get_args (src/utils/arguments.rs:46-58)
    for (const std::string &im : a.images) {
        const std::string ext = ".bmp";
        if (im.size() < ext.size() || im.compare(im.size() - ext.size(), ext.size(), ext) != 0)
            panic("Input image must be a .bmp file\n");
    }
    return a;
}

int algo_of(const Args &a) {
    return a.dct == "bin-dct" ? JPG_DCT_BIN : (a.dct == "real-dct-strict" ? JPG_DCT_REAL_STRICT : JPG_DCT_REAL);
}

std::string default_output(const std::string &image) { return image.substr(0, image.size() - 4) + ".jpeg"; }

[[noreturn]] void fail(jpg_ctx *ctx, int status) {
    std::string msg = jpg_strerror(status);
    if (status == JPG_ERR_HIP) msg += " (hipError " + std::to_string(jpg_last_hip_error(ctx)) + ")";
    if (ctx) jpg_destroy(ctx);
    panic(msg.c_str());
}

// One image cut into stripes: every stripe is transformed and coded by its own context (spread over the
// devices).  Each step is queued on all contexts before any of them is waited for, so the GPUs work at the
// same time; the few scalars that link neighbouring stripes pass through the host; the stuffed pieces are
// copied device to device (peer copies) into one buffer on the first device at their byte offsets.
// Returns the whole scan.
std::vector<uint8_t> encode_striped(const Args &args, const jph_bmp &bmp, int v, int algo) {
    const int n = args.stripes;
    std::vector<int32_t> y0(n), y1(n);
    if (jph_plan_stripes(bmp.height, v, n, y0.data(), y1.data()) != JPH_OK) panic("bad stripe plan");
    std::vector<jpg_ctx *> ctx(n, nullptr);
    void *gathered = nullptr;
    const int owner = args.devices[0];
    auto fail_all = [&](int k, int st) {
        jpg_ctx *bad = ctx[k];
        for (int j = 0; j < n; j++)
            if (ctx[j] && ctx[j] != bad) jpg_destroy(ctx[j]);
        if (gathered) jpg_device_free(owner, gathered);
        fail(bad, st);
    };
    std::vector<int> active;
    for (int k = 0; k < n; k++) {   // the copies of all stripes travel concurrently (one stream per context)
        if (y1[k] <= y0[k]) continue;
        int st = jpg_create(&ctx[k], args.devices[k % args.devices.size()], algo, args.ratio[0], args.ratio[1], args.ratio[2]);
        if (st != JPG_OK) fail_all(k, st);
        const uint8_t *rows = bmp.pixels + (size_t)(bmp.height - y1[k]) * bmp.row_stride;   // bottom-up file rows
        if ((st = jpg_stripe_host_async(ctx[k], rows, bmp.width, y1[k] - y0[k], bmp.row_stride, 1)) != JPG_OK) fail_all(k, st);
        active.push_back(k);
    }
    // The stripe chain where it applies (images at least one tile wide): a whole pass is queued on every context,
    // lowest rank first, and the kernels hand the linking scalars on themselves through a mailbox in page-locked
    // host memory; the pieces meet in a buffer on the first device when every device can write there, else in
    // page-locked host memory.  Anything it refuses (JPG_ERR_STATE: narrow image; JPG_ERR_CHAIN: scan larger than
    // the buffer) falls through to the steps below.
    if (n <= 64 && !active.empty() && active[0] == 0) {
        // (the buffer has two halves, one per pass in flight; the one pass here fills the first)
        const size_t cap = 2 * ((size_t)bmp.width * (size_t)bmp.height / 2 + (1u << 20));
        void *mailbox = jpg_chain_mailbox_alloc();
        bool peers = true;
        for (int k : active) peers = peers && jpg_peer_access(args.devices[k % args.devices.size()], owner) == 1;
        void *gather = peers ? jpg_device_alloc(owner, cap) : jpg_host_alloc(cap);
        uint64_t mask = 0;
        for (int k : active) mask |= 1ull << k;
        bool ok = mailbox && gather;
        for (size_t i = 0; ok && i < active.size(); i++) ok = jpg_chain_setup(ctx[active[i]], active[i], n, mailbox, gather, cap) == JPG_OK;
        size_t queued = 0;
        for (; ok && queued < active.size(); queued++) ok = jpg_chain_pass_async(ctx[active[queued]], mask) == JPG_OK;
        size_t total = 0;
        for (size_t i = queued; i-- > 0;) {   // everything queued is waited for, whatever happened
            size_t t = 0;
            const int st = jpg_chain_pass_finish(ctx[active[i]], nullptr, nullptr, nullptr, nullptr, &t);
            if (st != JPG_OK) ok = false;
            if (i == 0) total = t;
        }
        std::vector<uint8_t> scan;
        if (ok) {
            scan.resize(total);
            if (total && peers) ok = jpg_device_to_host(owner, scan.data(), gather, total) == JPG_OK;
            else if (total) std::memcpy(scan.data(), gather, total);
        }
        if (gather && peers) jpg_device_free(owner, gather);
        else if (gather) jpg_free(gather);
        if (mailbox) jpg_free(mailbox);
        if (ok) {
            for (int k = 0; k < n; k++)
                if (ctx[k]) jpg_destroy(ctx[k]);
            return scan;
        }
        // (the stripes are still loaded in their contexts: the steps below start from them)
        for (int k : active)
            if (int st = jpg_stripe_host_async(ctx[k], bmp.pixels + (size_t)(bmp.height - y1[k]) * bmp.row_stride, bmp.width,
                                               y1[k] - y0[k], bmp.row_stride, 1))
                fail_all(k, st);
    }
    // the DC values the following stripe predicts from: the last MCU of each stripe, transformed on its own
    std::vector<int16_t> dc(3 * (size_t)n, 0);
    for (int k : active)
        if (int st = jpg_last_dc_async(ctx[k])) fail_all(k, st);
    for (int k : active)
        if (int st = jpg_last_dc(ctx[k], &dc[3 * (size_t)k])) fail_all(k, st);
    std::vector<uint64_t> nbits(n, 0), offsets(n, 0);
    std::vector<uint32_t> tails(n, 0);
    std::vector<uint8_t> heads(n, 0);
    const int16_t zero[3] = {0, 0, 0};   // src/jpeg/entropy_coding.rs:38-40
    for (size_t i = 0; i < active.size(); i++) {
        const int k = active[i];
        const int16_t *pred = i ? &dc[3 * (size_t)active[i - 1]] : zero;
        if (int st = jpg_entropy_stripe_async(ctx[k], pred)) fail_all(k, st);
    }
    for (int k : active)
        if (int st = jpg_entropy_stripe_finish(ctx[k], &nbits[k], &tails[k])) fail_all(k, st);
    jph_stitch_plan(n, nbits.data(), tails.data(), offsets.data(), heads.data());
    const int last = active.empty() ? -1 : active.back();
    for (int k : active)
        if (int st = jpg_stitch_async(ctx[k], offsets[k], heads[k], k == last)) fail_all(k, st);
    std::vector<size_t> len(n, 0), at(n, 0);
    size_t total = 0;
    for (int k : active) {
        uint8_t tail = 0;
        if (int st = jpg_stitch_finish(ctx[k], &len[k], &tail)) fail_all(k, st);
        at[k] = total;
        total += len[k];
    }
    std::vector<uint8_t> scan(total);
    if (total) {
        gathered = jpg_device_alloc(owner, total);
        if (!gathered) panic("out of device memory for the gathered scan");
        for (int k : active)
            if (int st = jpg_scan_copy_to_device(ctx[k], static_cast<uint8_t *>(gathered) + at[k], owner, len[k])) fail_all(k, st);
        for (int k : active)
            if (int st = jpg_sync(ctx[k])) fail_all(k, st);
        if (int st = jpg_device_to_host(owner, scan.data(), gathered, total)) fail_all(active[0], st);
        jpg_device_free(owner, gathered);
    }
    for (int k = 0; k < n; k++)
        if (ctx[k]) jpg_destroy(ctx[k]);
    return scan;
}

void encode_one(const Args &args, const std::string &image, const std::string &output, int device) {
    std::printf("Loading bmp...\n");
    std::fflush(stdout);
    jph_bmp bmp{};
    // the pixel array goes straight into page-locked memory, so the device reads it by DMA
    const int be = jph_bmp_read_with(image.c_str(), &bmp, jpg_host_alloc, jpg_free);
    if (be == JPH_ERR_OPEN) panic("Could not open bitmap image file!\n");
    if (be == JPH_ERR_FORMAT)
        panic("Unsupported bitmap: only uncompressed 24-bit bottom-up files with positive dimensions are read\n");
    if (be != JPH_OK) panic("could not read the bitmap image file");
    const int h = 1 + (args.ratio[1] == 2), v = 1 + (args.ratio[1] == 2 && args.ratio[2] == 0);
    if (args.stripes > 0) {
        const std::vector<uint8_t> scan = encode_striped(args, bmp, v, algo_of(args));
        std::printf("Loaded!\nChrominance downsampling...\nDone!\n");
        std::printf("Performing Discrete Cosine Transform + Quantization...\nDone!\nEntropy encoding...\nDone!\n");
        std::printf("Creating file\n");
        std::fflush(stdout);
        const int we = jph_jpeg_write(output.c_str(), bmp.width, bmp.height, h, v, scan.data(), scan.size());
        jpg_free(bmp.pixels);
        if (we != JPH_OK) panic("called `Result::unwrap()` on an `Err` value: could not write the output file");
        std::printf("Done!\n");
        return;
    }
    jpg_ctx *ctx = nullptr;
    int st = jpg_create(&ctx, device, algo_of(args), args.ratio[0], args.ratio[1], args.ratio[2]);
    if (st != JPG_OK) fail(ctx, st);

    uint8_t *scan = nullptr;
    size_t scan_len = 0;
    std::vector<uint8_t> staged_scan;
    uint8_t header[1024];
    size_t header_len = 0;
    if (args.fused) {
        // The encode is queued (rows copied and coded band by band) and the file's header is put together while
        // the device works; header, scan and end marker then leave in one write.
        st = jpg_encode_scan_begin(ctx, bmp.pixels, bmp.width, bmp.height, bmp.row_stride, 1);
        if (st != JPG_OK) fail(ctx, st);
        header_len = jph_jfif_header(header, sizeof header, bmp.width, bmp.height, h, v);
        st = jpg_encode_scan_end(ctx, &scan, &scan_len);
        if (st != JPG_OK) fail(ctx, st);
        std::printf("Loaded!\nChrominance downsampling...\nDone!\n");
        std::printf("Performing Discrete Cosine Transform + Quantization...\nDone!\nEntropy encoding...\nDone!\n");
    } else {
        // the five calls of main (src/main.rs:17,35,48,61,66)
        if ((st = jpg_load_bgr(ctx, bmp.pixels, bmp.width, bmp.height, bmp.row_stride, 1)) != JPG_OK) fail(ctx, st);
        std::printf("Loaded!\n");
        std::printf("Chrominance downsampling...\n");
        if ((st = jpg_downsample(ctx)) != JPG_OK) fail(ctx, st);
        std::printf("Done!\n");
        std::printf("Performing Discrete Cosine Transform + Quantization...\n");
        if ((st = jpg_dct_quant(ctx)) != JPG_OK) fail(ctx, st);
        std::printf("Done!\n");
        std::printf("Entropy encoding...\n");
        uint64_t nbits = 0;
        if ((st = jpg_entropy(ctx, &nbits)) != JPG_OK) fail(ctx, st);
        std::printf("Done!\n");
    }
    std::printf("Creating file\n");
    std::fflush(stdout);
    if (!args.fused) {
        if ((st = jpg_stuff(ctx, &scan_len)) != JPG_OK) fail(ctx, st);
        staged_scan.resize(scan_len ? scan_len : 1);
        if ((st = jpg_get_scan(ctx, staged_scan.data(), staged_scan.size())) != JPG_OK) fail(ctx, st);
    }
    int we = JPH_OK;
    if (args.fused) {
        static const uint8_t eoi[2] = {0xFF, 0xD9};   // src/jpeg/file.rs:134
        const int fd = ::open(output.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || header_len == 0) {
            we = JPH_ERR_IO;
        } else {
            iovec part[3] = {{header, header_len}, {scan, scan_len}, {const_cast<uint8_t *>(eoi), sizeof eoi}};
            size_t left = header_len + scan_len + sizeof eoi;
            int k = 0;
            while (left && we == JPH_OK) {             // (a write may stop short of the whole)
                const ssize_t n = ::writev(fd, part + k, 3 - k);
                if (n <= 0) { we = JPH_ERR_IO; break; }
                left -= (size_t)n;
                size_t done = (size_t)n;
                while (k < 3 && done >= part[k].iov_len) done -= part[k++].iov_len;
                if (k < 3) { part[k].iov_base = static_cast<uint8_t *>(part[k].iov_base) + done; part[k].iov_len -= done; }
            }
        }
        if (fd >= 0 && ::close(fd) != 0) we = JPH_ERR_IO;
    } else {
        we = jph_jpeg_write(output.c_str(), bmp.width, bmp.height, h, v, staged_scan.data(), scan_len);
    }
    if (scan) jpg_free(scan);
    jpg_destroy(ctx);
    jpg_free(bmp.pixels);
    if (we != JPH_OK) panic("called `Result::unwrap()` on an `Err` value: could not write the output file");
    std::printf("Done!\n");
}

}  // namespace

int main(int argc, char **argv) {
    const Args args = parse(argc, argv);
    for (size_t k = 0; k < args.images.size(); k++) {
        const std::string &image = args.images[k];
        const std::string output = args.output.empty() ? default_output(image) : args.output;
        // Args::print_args (src/utils/arguments.rs:60-66)
        std::printf("image: \"%s\"\n", image.c_str());
        std::printf("output: \"%s\"\n", output.c_str());
        std::printf("subsampling ratio: \"(%d, %d, %d)\"\n", args.ratio[0], args.ratio[1], args.ratio[2]);
        std::printf("dct algorithm: \"%s\"\n", args.dct == "bin-dct" ? "BinDct" : "RealDct");
        std::printf("\n");
        encode_one(args, image, output, args.devices[k % args.devices.size()]);
    }
    return 0;
}

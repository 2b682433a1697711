/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker.  The product (hypergrep_amd/) never links, imports or calls this code.
 *
 * orx: a deliberately simple CPU restatement of the regex semantics the reference obtains
 * from Intel Hyperscan 5.4.2 (third-party, absent from /root/reference; version pinned at
 * /root/reference/utils/build_hyperscanner.sh:9,49).  Call sites being restated:
 * hs_compile_multi at hypergrep/lib/c/hyperscanner.c:136 and block-mode hs_scan at :217.
 *
 * Algorithm: recursive-descent parser for the PCRE subset Hyperscan documents as supported
 * (byte semantics, no UTF-8/UCP), Thompson program, Pike-style set simulation that reports
 * every distinct match END offset in ascending order (Hyperscan's "all matches, by end
 * offset" semantics), or only the first when HS_FLAG_SINGLEMATCH is set.
 *
 * Parity status: regex *semantics* are pinned by (a) every engine-touching expectation in the
 * reference's own test tables (hypergrep/test/test_hypergrep.py:64-74,161-290,292-909, replayed
 * through the reference's unmodified shim + Python in tests/golden/make_golden.py) and (b) a
 * Python `re`-on-bytes cross-check.  Constructs the reference never tests (classes, anchors,
 * \b, bounded ranges, distinct ids) are "parity unpinned" against real Hyperscan.
 */
#ifndef ORX_H
#define ORX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flag values from hypergrep/utils.py:10-13 (hs_compile.h). */
#define ORX_FLAG_CASELESS 1u
#define ORX_FLAG_DOTALL 2u
#define ORX_FLAG_MULTILINE 4u
#define ORX_FLAG_SINGLEMATCH 8u
#define ORX_FLAGS_SUPPORTED 15u

typedef struct orx_prog orx_prog;

/* Compile one expression.  Returns NULL and fills err on rejection (anything Hyperscan
 * documents as unsupported, unsupported flags, or a pattern that can match the empty string). */
orx_prog *orx_compile(const char *pattern, unsigned flags, char *err, size_t errlen);
void orx_free(orx_prog *p);

/* Report distinct match end offsets of p in data[0,len) in ascending order.  cb returning
 * non-zero stops the scan.  `single` stops after the first report.  Returns number of reports. */
typedef int (*orx_report_fn)(size_t to, void *ctx);
size_t orx_scan(const orx_prog *p, const unsigned char *data, size_t len, int single, orx_report_fn cb, void *ctx);

#ifdef __cplusplus
}
#endif
#endif

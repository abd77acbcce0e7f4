/*
 * reccut.h -- C ABI of libreccut.so: in-process protein-domain segmentation.
 *
 * Replaces the reference's external helper: mgtools/DCTdomain src/RecCut.cpp (a
 * stand-alone executable, 463 lines) and its calling protocol in
 * src/fingerprint.py:83-107 (write a .ce text file, spawn the binary, parse stdout).
 * Same integer recurrences, same thresholds, same output strings -- without the file
 * and process round trip.  Host-side C++ (the algorithm is an O(V^2) integer scan with
 * data-dependent recursion; it is not GPU work).
 */
#ifndef RECCUT_H
#define RECCUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECCUT_OK 0
#define RECCUT_ERR_INVALID (-1)  /* bad argument / contact index outside the protein */
#define RECCUT_ERR_BUFFER (-2)   /* output buffer too small */
#define RECCUT_ERR_UNDEFINED (-3) /* the reference binary would index outside its segment table here
                                     (undefined behaviour in src/RecCut.cpp:16-148); no answer is defined */

/* Thresholds of src/RecCut.cpp:10-14 (overridable there by --cutoff a b). */
#define RECCUT_CUT1_DEFAULT 0.08
#define RECCUT_CUT2_DEFAULT 0.07

/* One protein.  Contacts are the "CON" entries of the .ce file (src/fingerprint.py:69-74):
 * 0-based residue pairs with the probability AS PRINTED THERE ("%.6f" of the float32 value,
 * which this function reproduces from `prob`).  Graph construction = readGraph
 * (src/RecCut.cpp:354-397): weight (int)(p*100+0.5), symmetric, then |i-j| <= 3 forced to 100.
 * Writes the domain list exactly as the binary prints it after "<name> <count> ":
 *     "1-70,180-261;71-179;"   (1-based, discontinuous pieces joined by ',', every domain ends with ';')
 * into out (NUL-terminated) and the number of domains into n_domains.
 * A protein shorter than 22 residues is one domain "1-L;" (src/RecCut.cpp:446-447). */
int reccut_predict(int32_t n_res, const int32_t* ci, const int32_t* cj, const float* prob, int64_t n_contacts,
                   double cut1, double cut2, char* out, int64_t out_cap, int32_t* n_domains);

/* The integer edge weight the reference derives from a contact probability: the float32 printed as "%.6f" into the
 * .ce file (src/fingerprint.py:72), read back by the binary and turned into (int)(v*100+0.5) (src/RecCut.cpp:384).
 * Computed without the text round trip unless the value sits within 1e-5 of a rounding boundary. */
int32_t reccut_contact_weight(float prob);

/* The same weight in exact arithmetic, without the text round trip (p * 10^6 is exact in double; "%.6f" rounds that exact value
 * half-to-even; strtod returns the double nearest to n / 10^6): what the GPU cutter (dctfp_reccut, include/dctfp.h) computes.
 * tests/test_reccut.py holds the two against each other. */
int32_t reccut_contact_weight_exact(float prob);

/* The strings of dctfp_reccut's encoded results (include/dctfp.h: {D, then per domain n_segs, (first, last) ...} at enc +
 * enc_off[p]), packed like reccut_predict_packed's: protein p's "1-70,180-261;71-179;" at out[out_off[p], out_off[p+1]).
 * needs_host[p] = 1 (and an empty string) where the GPU left the protein to this library (status -1). */
int reccut_format_packed(int64_t n_prot, const int32_t* enc, const int64_t* enc_off, char* out, int64_t out_cap, int64_t* out_off,
                         int32_t* n_domains, uint8_t* needs_host);

/* Many proteins on n_threads host threads.  Protein p uses contacts [offs[p], offs[p+1]) and writes
 * its string at out + p * out_stride.  rc[p] receives the per-protein return code. */
int reccut_predict_batch(int64_t n_prot, const int32_t* n_res, const int64_t* offs, const int32_t* ci,
                         const int32_t* cj, const float* prob, double cut1, double cut2, char* out,
                         int64_t out_stride, int32_t* n_domains, int32_t* rc, int32_t n_threads);

/* The same with the strings packed back to back: protein p's string (no NUL) is out[out_off[p], out_off[p+1]); out_off has
 * n_prot + 1 entries.  Proteins are handed to the threads one at a time (long ones do not queue up behind each other).
 * What a database flush uses: no n_prot x stride scratch to zero and to search for terminators. */
int reccut_predict_packed(int64_t n_prot, const int32_t* n_res, const int64_t* offs, const int32_t* ci, const int32_t* cj,
                          const float* prob, double cut1, double cut2, char* out, int64_t out_cap, int64_t* out_off,
                          int32_t* n_domains, int32_t* rc, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif

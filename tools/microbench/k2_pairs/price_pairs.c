/*
 * tools/microbench/k2_pairs/price_pairs.c -- prices a K2 lookup whose entries carry the extended value and, where two short
 * symbols fit the index, BOTH of them ("Next round" 1), on real streams, on the CPU, before any kernel is built.
 *
 * K2 (k2_huffman.hip) decodes 64 restart intervals per wave in LOCK-STEP: block b of every lane's MCU m is decoded together, the
 * wave iterates until its slowest lane has finished the block.  The cost of a wave-block is therefore
 *      max over the 64 lanes of (symbol steps of the lane's block)
 * and a lookup that delivers two symbols in one step only pays where it shortens that maximum.  This tool decodes a baseline
 * file (one interleaved scan, DRI > 0; its own small Huffman decoder: test tooling, not the product and not the oracle), records
 * every block's symbols, and replays the wave schedule under:
 *   single      today's loop: one step per symbol (DC, every AC symbol incl. EOB / ZRL)
 *   pair LB     an LB-bit first level whose entry holds two AC symbols when code+magnitude of both fit LB bits, the first is not
 *               EOB and does not end the block, and the first value fits 5 signed bits (the 32-bit entry layout of DESIGN 3 K2)
 *   triple LB   the same with up to three (an upper bound for wider entries)
 * and reports wave-steps (the time), lane-steps (the work), and how often a wave-step has at least one lane on the paths that
 * leave the fast path: "medium" (a single symbol whose code fits the index but whose magnitude does not: the value comes from
 * the stream) and "long" (code longer than the index).
 *
 * build: gcc -O2 -o price_pairs price_pairs.c ; run: ./price_pairs file.jpg [file.jpg ...]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint8_t bits[17]; uint8_t vals[256]; int mincode[18], maxcode[18], valptr[18]; } htab;
static void build(htab *h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->bits[l];
        k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
}
typedef struct { const uint8_t *p, *end; uint64_t acc; int n; } br;
static void fill(br *b) {
    while (b->n <= 56) {
        int c = 0xFF;  /* ones behind the data */
        if (b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0) b->p += 2;
                else c = 0xFF, b->p = b->end;  /* a marker: the interval is over */
            } else b->p++;
        }
        b->acc |= (uint64_t)c << (56 - b->n);
        b->n += 8;
    }
}
static int getbits(br *b, int n) { if (!n) return 0; fill(b); int v = (int)(b->acc >> (64 - n)); b->acc <<= n; b->n -= n; return v; }
static int decode(br *b, const htab *h, int *len) {
    fill(b);
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (int)(b->acc >> (64 - l));
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) {
            b->acc <<= l; b->n -= l; *len = l;
            return h->vals[h->valptr[l] + code - h->mincode[l]];
        }
    }
    fprintf(stderr, "bad code\n"); exit(2);
}
/* one symbol of a block: total bits (code + magnitude), code bits, advance (63 = EOB), |value| */
typedef struct { uint8_t n, code, adv; int16_t val; } sym;

typedef struct { uint64_t wave_steps, lane_steps, medium_steps, long_steps, blocks; } tally;

#define MAXSYM 66
static int steps_of(const sym *s, int ns, int mode, int lb, int *medium, int *lng) {
    /* s[0] = DC, s[1..] = AC; returns the lane's step count for this block; flags whether any step left the fast path */
    int steps = 1;  /* DC */
    if (s[0].code > 11) *lng = 1;
    int i = 1, pos = 1;
    while (i < ns) {
        const sym *a = &s[i];
        int take = 1;
        if (mode >= 2 && a->code <= lb) {
            int used = a->n, p = pos + a->adv;
            if (a->adv != 63 && a->n <= lb && a->val >= -16 && a->val <= 15 && p < 64) {
                /* a second (and third) symbol of the SAME block that still fits */
                while (take < mode - 1 + 1 && i + take < ns) {
                    const sym *c = &s[i + take];
                    if (used + c->n > lb) break;
                    if (take + 1 < mode && !(c->val >= -16 && c->val <= 15)) { /* only the LAST value may be wide */ }
                    used += c->n;
                    take++;
                    if (c->adv == 63) break;
                    p += c->adv;
                    if (p >= 64) break;
                    if (take >= mode) break;
                }
            }
        }
        if (a->code > lb) *lng = 1;
        else if (take == 1 && a->n > lb) *medium = 1;
        for (int t = 0; t < take; t++) pos += s[i + t].adv;
        i += take;
        steps++;
    }
    return steps;
}

int main(int argc, char **argv) {
    tally T[8];
    memset(T, 0, sizeof T);
    static const int modes[8][2] = {{1, 11}, {2, 10}, {2, 11}, {2, 12}, {3, 11}, {3, 12}, {2, 13}, {3, 16}};
    uint64_t nsym_ac = 0, nsym_total = 0, nblocks = 0, hist_n[40] = {0};
    for (int f = 1; f < argc; f++) {
        FILE *fp = fopen(argv[f], "rb");
        if (!fp) { perror(argv[f]); return 1; }
        fseek(fp, 0, SEEK_END); long len = ftell(fp); fseek(fp, 0, SEEK_SET);
        uint8_t *d = malloc(len + 16); if (fread(d, 1, len, fp) != (size_t)len) return 1; fclose(fp);
        htab dc[4], ac[4]; memset(dc, 0, sizeof dc); memset(ac, 0, sizeof ac);
        int W = 0, H = 0, nc = 0, ch[4], cv[4], dri = 0, td[4], ta[4];
        long p = 2, sos = -1;
        while (p + 4 <= len) {
            int m = d[p + 1], l = (d[p + 2] << 8) | d[p + 3];
            if (m == 0xC0) { H = (d[p + 5] << 8) | d[p + 6]; W = (d[p + 7] << 8) | d[p + 8]; nc = d[p + 9];
                for (int c = 0; c < nc; c++) ch[c] = d[p + 11 + 3 * c] >> 4, cv[c] = d[p + 11 + 3 * c] & 15; }
            else if (m == 0xC4) { long q = p + 4; while (q < p + 2 + l) { htab *h = (d[q] >> 4) ? &ac[d[q] & 3] : &dc[d[q] & 3]; int n = 0;
                    for (int i = 1; i <= 16; i++) { h->bits[i] = d[q + i]; n += d[q + i]; }
                    memcpy(h->vals, d + q + 17, n); build(h); q += 17 + n; } }
            else if (m == 0xDD) dri = (d[p + 4] << 8) | d[p + 5];
            else if (m == 0xDA) { for (int c = 0; c < nc; c++) td[c] = d[p + 6 + 2 * c] >> 4, ta[c] = d[p + 6 + 2 * c] & 15; sos = p + 2 + l; break; }
            p += 2 + l;
        }
        if (sos < 0 || dri == 0) { fprintf(stderr, "%s: need one scan with DRI > 0\n", argv[f]); return 1; }
        int hmax = 1, vmax = 1, bpm = 0;
        for (int c = 0; c < nc; c++) { if (ch[c] > hmax) hmax = ch[c]; if (cv[c] > vmax) vmax = cv[c]; bpm += ch[c] * cv[c]; }
        long mcus = (long)((W + 8 * hmax - 1) / (8 * hmax)) * ((H + 8 * vmax - 1) / (8 * vmax));
        long nint = (mcus + dri - 1) / dri;
        /* per interval: per (mcu, block): symbol list.  Kept per WAVE of 64 intervals to bound memory. */
        sym (*blk)[MAXSYM] = malloc(sizeof(sym[MAXSYM]) * 64 * dri * bpm);
        uint8_t *nsy = malloc(64 * dri * bpm);
        const uint8_t *q = d + sos, *end = d + len;
        for (long w0 = 0; w0 < nint; w0 += 64) {
            int lanes = nint - w0 < 64 ? (int)(nint - w0) : 64;
            memset(nsy, 0, 64 * dri * bpm);
            for (int ln = 0; ln < lanes; ln++) {
                long iv = w0 + ln;
                /* the interval's bytes: up to the next marker */
                const uint8_t *e = q;
                while (e + 1 < end && !(e[0] == 0xFF && e[1] != 0 )) e++;
                br b = {q, e, 0, 0};
                long m_here = (iv == nint - 1) ? mcus - iv * dri : dri;
                for (long m = 0; m < m_here; m++) {
                    int bi = 0;
                    for (int c = 0; c < nc; c++)
                        for (int k = 0; k < ch[c] * cv[c]; k++, bi++) {
                            sym *s = blk[(ln * dri + m) * bpm + bi];
                            int ns = 0, cl;
                            int t = decode(&b, &dc[td[c]], &cl);
                            int v = getbits(&b, t);
                            s[ns++] = (sym){(uint8_t)(cl + t), (uint8_t)cl, 0, 0};
                            (void)v;
                            int k2 = 1;
                            while (k2 < 64) {
                                int rs = decode(&b, &ac[ta[c]], &cl), r = rs >> 4, sz = rs & 15;
                                int raw = getbits(&b, sz);
                                int val = sz ? (raw < (1 << (sz - 1)) ? raw - (1 << sz) + 1 : raw) : 0;
                                int adv = sz ? r + 1 : (r ? 16 : 63);
                                s[ns++] = (sym){(uint8_t)(cl + sz), (uint8_t)cl, (uint8_t)adv, (int16_t)val};
                                nsym_ac++;
                                hist_n[cl + sz]++;
                                if (adv == 63) break;
                                k2 += adv;
                            }
                            nsym_total += ns;
                            nblocks++;
                            nsy[(ln * dri + m) * bpm + bi] = (uint8_t)ns;
                        }
                }
                q = e + 2;  /* behind the RSTn */
            }
            for (int md = 0; md < 8; md++)
                for (long m = 0; m < dri; m++)
                    for (int bi = 0; bi < bpm; bi++) {
                        int mx = 0, medium = 0, lng = 0; uint64_t sum = 0; int any = 0;
                        for (int ln = 0; ln < lanes; ln++) {
                            int ns = nsy[(ln * dri + m) * bpm + bi];
                            if (!ns) continue;
                            any = 1;
                            int st = steps_of(blk[(ln * dri + m) * bpm + bi], ns, modes[md][0], modes[md][1], &medium, &lng);
                            sum += st;
                            if (st > mx) mx = st;
                        }
                        if (!any) continue;
                        T[md].wave_steps += mx; T[md].lane_steps += sum; T[md].blocks++;
                        T[md].medium_steps += medium; T[md].long_steps += lng;
                    }
        }
        free(blk); free(nsy); free(d);
    }
    uint64_t bits_ac = 0;
    for (int n = 1; n < 40; n++) bits_ac += hist_n[n] * (uint64_t)n;
    printf("blocks %llu, symbols per block %.2f, mean bits per AC symbol (code + magnitude) %.2f\n", (unsigned long long)nblocks, (double)nsym_total / nblocks,
           (double)bits_ac / nsym_ac);
    uint64_t cum = 0;
    printf("AC symbol length (code + magnitude) cumulative share: ");
    for (int n = 1; n < 33; n++) { cum += hist_n[n]; if (n >= 8 && n <= 16) printf("<=%d: %.4f  ", n, (double)cum / nsym_ac); }
    printf("\n%-12s %14s %10s %12s %10s %14s %14s\n", "mode", "wave-steps", "vs single", "lane-steps", "lock-step", "wave-blocks w/", "wave-blocks w/");
    printf("%-12s %14s %10s %12s %10s %14s %14s\n", "", "per wave-block", "", "per block", "idle", "a medium lane", "a long code");
    for (int md = 0; md < 8; md++) {
        char name[32];
        snprintf(name, sizeof name, "%s %d", modes[md][0] == 1 ? "single" : (modes[md][0] == 2 ? "pair" : "triple"), modes[md][1]);
        printf("%-12s %14.3f %10.3f %12.3f %9.1f%% %13.1f%% %13.1f%%\n", name, (double)T[md].wave_steps / T[md].blocks,
               (double)T[md].wave_steps / T[0].wave_steps, (double)T[md].lane_steps / nblocks,
               100.0 * (1.0 - (double)T[md].lane_steps / (64.0 * T[md].wave_steps)), 100.0 * T[md].medium_steps / T[md].blocks, 100.0 * T[md].long_steps / T[md].blocks);
    }
    return 0;
}

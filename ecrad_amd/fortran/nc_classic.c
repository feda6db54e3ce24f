/* nc_classic.c -- netCDF "classic" files (CDF-1, and CDF-2 = 64-bit offsets) read and written from scratch.
 *
 * The Fortran host of the radiation() drop-in (SURVEY.md 8(b) last row: "netCDF reader/writer ... re-written in the repo")
 * does its file I/O through the `netcdf` module of netcdf.F90, which is a thin generic-interface layer over this file.
 * Every input, look-up-table and golden file of the reference is classic format (magic "CDF\001"), and the reference's
 * own wrapper utilities/easy_netcdf.F90 uses 25 entry points of the nf90 API; those are what exists here.
 *
 * File format (NetCDF Users Guide, "File Format Specification", classic):
 *   header   = magic numrecs dim_list gatt_list var_list        all integers big-endian
 *   x_list   = ABSENT(0,0) | tag nelems [x ...]                 tags: dimension 0x0A, variable 0x0B, attribute 0x0C
 *   dim      = name length                                      length 0 = the record dimension
 *   attr     = name nc_type nelems [values, padded to 4 bytes]
 *   var      = name rank [dimid ...] vatt_list nc_type vsize begin   (begin: 4 bytes in CDF-1, 8 in CDF-2)
 *   data     = fixed-size variables in order, then the records: each record holds one slab of every record variable
 * Numeric values are converted between the file's type and the caller's memory type (as nc_get_vara_double etc. do).
 *
 * Scope: everything is synchronous stdio; define mode -> nc_enddef -> data mode, no re-entering define mode; files being
 * written have fixed dimensions only (the reference never defines NF90_UNLIMITED); record variables are READ (the
 * reference's test inputs have "column" as record dimension).  Not thread-safe per file; the table of open files is
 * guarded by the caller (Fortran I/O in the reference's driver is serial).
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ECNC_MAX_FILES 64
#define ECNC_MAX_DIMS 8
#define ECNC_MAX_NAME 256

enum { T_BYTE = 1, T_CHAR = 2, T_SHORT = 3, T_INT = 4, T_FLOAT = 5, T_DOUBLE = 6 };
enum { E_BADID = -33, E_NFILE = -34, E_INVAL = -36, E_PERM = -37, E_NOTINDEFINE = -38, E_INDEFINE = -39, E_INVALCOORDS = -40,
       E_MAXDIMS = -41, E_NAMEINUSE = -42, E_NOTATT = -43, E_BADTYPE = -45, E_BADDIM = -46, E_NOTVAR = -49, E_NOTNC = -51,
       E_MAXNAME = -53, E_UNLIMIT = -54, E_CHAR = -56, E_EDGE = -57, E_RANGE = -60, E_NOMEM = -61, E_HDFERR = -101, E_NOHDF5 = -1101 };

typedef struct { char name[ECNC_MAX_NAME]; uint64_t len; } Dim;
typedef struct { char name[ECNC_MAX_NAME]; int type; uint64_t n; void* v; /* n values of the file type, host byte order */ } Att;
typedef struct {
  char name[ECNC_MAX_NAME];
  int rank, dimid[ECNC_MAX_DIMS], type, natt, is_rec;
  Att* att;
  uint64_t vsize, begin;
} Var;
typedef struct {
  FILE* fp;
  int64_t h5;      /* version 6: the file is netCDF-4 / HDF5 and is READ through the HDF5 library (ecnc_h5_open) */
  int used, writable, define_mode, version;
  uint64_t numrecs, recsize;
  int ndim, nvar, ngatt, recdim;
  Dim* dim;
  Var* var;
  Att* gatt;
} File;

static File g_files[ECNC_MAX_FILES];
static int ecnc_h5_enddef(File* f);
static int ecnc_h5_open(File* f, const char* path);
static int ecnc_h5_read(File* f, const Var* v, int memtype, void* buf, const long long* s, const long long* c);
static void ecnc_h5_close(File* f);

static size_t tsize(int t) { return t == T_BYTE || t == T_CHAR ? 1 : t == T_SHORT ? 2 : t == T_INT || t == T_FLOAT ? 4 : t == T_DOUBLE ? 8 : 0; }
static uint64_t pad4(uint64_t n) { return (n + 3) & ~(uint64_t)3; }
static File* file_of(int ncid) {
  const int i = ncid - 1000;
  return (i >= 0 && i < ECNC_MAX_FILES && g_files[i].used) ? &g_files[i] : NULL;
}

/* ---- big-endian primitives ------------------------------------------------------------------------ */
static int rd_u32(FILE* fp, uint32_t* v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) return E_NOTNC; *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; return 0; }
static int rd_u64(FILE* fp, uint64_t* v) { uint32_t a, b; if (rd_u32(fp, &a) || rd_u32(fp, &b)) return E_NOTNC; *v = ((uint64_t)a << 32) | b; return 0; }
static void wr_u32(FILE* fp, uint32_t v) { unsigned char b[4] = {(unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v}; fwrite(b, 1, 4, fp); }
static void wr_u64(FILE* fp, uint64_t v) { wr_u32(fp, (uint32_t)(v >> 32)); wr_u32(fp, (uint32_t)v); }
static void swap_in_place(void* p, size_t ts, uint64_t n) {
  unsigned char* b = (unsigned char*)p;
  if (ts == 1) return;
  for (uint64_t i = 0; i < n; ++i, b += ts)
    for (size_t k = 0; k < ts / 2; ++k) { unsigned char t = b[k]; b[k] = b[ts - 1 - k]; b[ts - 1 - k] = t; }
}

static int rd_name(FILE* fp, char* name) {
  uint32_t n;
  if (rd_u32(fp, &n)) return E_NOTNC;
  if (n >= ECNC_MAX_NAME) return E_MAXNAME;
  char buf[ECNC_MAX_NAME + 4];
  if (fread(buf, 1, pad4(n), fp) != pad4(n)) return E_NOTNC;
  memcpy(name, buf, n);
  name[n] = 0;
  return 0;
}
static void wr_name(FILE* fp, const char* name) {
  const uint32_t n = (uint32_t)strlen(name);
  const char zero[4] = {0, 0, 0, 0};
  wr_u32(fp, n);
  fwrite(name, 1, n, fp);
  fwrite(zero, 1, pad4(n) - n, fp);
}
static uint64_t name_bytes(const char* name) { return 4 + pad4(strlen(name)); }

static int rd_atts(FILE* fp, int* natt, Att** att) {
  uint32_t tag, n;
  if (rd_u32(fp, &tag) || rd_u32(fp, &n)) return E_NOTNC;
  *natt = 0; *att = NULL;
  if (tag == 0 && n == 0) return 0;
  if (tag != 0x0C) return E_NOTNC;
  *att = (Att*)calloc(n ? n : 1, sizeof(Att));
  if (!*att) return E_NOMEM;
  *natt = (int)n;
  for (uint32_t i = 0; i < n; ++i) {
    Att* a = &(*att)[i];
    uint32_t t, ne;
    int st = rd_name(fp, a->name);
    if (st) return st;
    if (rd_u32(fp, &t) || rd_u32(fp, &ne)) return E_NOTNC;
    if (!tsize((int)t)) return E_BADTYPE;
    a->type = (int)t; a->n = ne;
    const uint64_t nb = (uint64_t)ne * tsize(a->type);
    a->v = malloc(pad4(nb) + 8);
    if (!a->v) return E_NOMEM;
    if (fread(a->v, 1, pad4(nb), fp) != pad4(nb)) return E_NOTNC;
    swap_in_place(a->v, tsize(a->type), ne);
  }
  return 0;
}
static uint64_t atts_bytes(int natt, const Att* att) {
  uint64_t b = 8;
  for (int i = 0; i < natt; ++i) b += name_bytes(att[i].name) + 8 + pad4(att[i].n * tsize(att[i].type));
  return b;
}
static void wr_atts(FILE* fp, int natt, const Att* att) {
  if (natt == 0) { wr_u32(fp, 0); wr_u32(fp, 0); return; }
  wr_u32(fp, 0x0C); wr_u32(fp, (uint32_t)natt);
  for (int i = 0; i < natt; ++i) {
    const Att* a = &att[i];
    const size_t ts = tsize(a->type);
    const uint64_t nb = a->n * ts;
    wr_name(fp, a->name);
    wr_u32(fp, (uint32_t)a->type); wr_u32(fp, (uint32_t)a->n);
    unsigned char* tmp = (unsigned char*)calloc(1, pad4(nb) + 8);
    memcpy(tmp, a->v, nb);
    swap_in_place(tmp, ts, a->n);
    fwrite(tmp, 1, pad4(nb), fp);
    free(tmp);
  }
}

static void free_file(File* f) {
  for (int i = 0; i < f->ngatt; ++i) free(f->gatt[i].v);
  for (int v = 0; v < f->nvar; ++v) {
    for (int i = 0; i < f->var[v].natt; ++i) free(f->var[v].att[i].v);
    free(f->var[v].att);
  }
  free(f->gatt); free(f->var); free(f->dim);
  if (f->version == 6) ecnc_h5_close(f);
  if (f->fp) fclose(f->fp);
  memset(f, 0, sizeof(*f));
}

static int new_slot(File** out, int* ncid) {
  for (int i = 0; i < ECNC_MAX_FILES; ++i)
    if (!g_files[i].used) {
      memset(&g_files[i], 0, sizeof(File));
      g_files[i].used = 1; g_files[i].recdim = -1;
      *out = &g_files[i]; *ncid = 1000 + i;
      return 0;
    }
  return E_NFILE;
}

/* elements of one slab of a variable (the record dimension counts as 1) */
static uint64_t var_slab_elems(const File* f, const Var* v) {
  uint64_t n = 1;
  for (int k = 0; k < v->rank; ++k) if (!(k == 0 && v->is_rec)) n *= f->dim[v->dimid[k]].len;
  return n;
}

/* ---- open ------------------------------------------------------------------------------------------ */
int ecnc_open(const char* path, int* ncid) {
  File* f;
  int st = new_slot(&f, ncid);
  if (st) return st;
  f->fp = fopen(path, "rb");
  if (!f->fp) { st = errno ? errno : 2; f->used = 0; return st; }
  unsigned char magic[4];
  uint32_t tag, n, u;
  if (fread(magic, 1, 4, f->fp) != 4) { free_file(f); return E_NOTNC; }
  if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') {      /* netCDF-4: an HDF5 file (signature \211HDF\r\n\032\n) */
    fclose(f->fp); f->fp = NULL;
    st = ecnc_h5_open(f, path);
    if (st) free_file(f);
    return st;
  }
  if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2)) { free_file(f); return E_NOTNC; }
  f->version = magic[3];
  if (rd_u32(f->fp, &u)) { free_file(f); return E_NOTNC; }
  f->numrecs = u;
  if (rd_u32(f->fp, &tag) || rd_u32(f->fp, &n)) { free_file(f); return E_NOTNC; }
  if (!(tag == 0 && n == 0) && tag != 0x0A) { free_file(f); return E_NOTNC; }
  f->ndim = (int)n;
  f->dim = (Dim*)calloc(n ? n : 1, sizeof(Dim));
  for (uint32_t i = 0; i < n; ++i) {
    if ((st = rd_name(f->fp, f->dim[i].name)) || rd_u32(f->fp, &u)) { free_file(f); return st ? st : E_NOTNC; }
    f->dim[i].len = u;
    if (u == 0) f->recdim = (int)i;
  }
  if ((st = rd_atts(f->fp, &f->ngatt, &f->gatt))) { free_file(f); return st; }
  if (rd_u32(f->fp, &tag) || rd_u32(f->fp, &n)) { free_file(f); return E_NOTNC; }
  if (!(tag == 0 && n == 0) && tag != 0x0B) { free_file(f); return E_NOTNC; }
  f->nvar = (int)n;
  f->var = (Var*)calloc(n ? n : 1, sizeof(Var));
  int nrec = 0;
  for (uint32_t i = 0; i < n; ++i) {
    Var* v = &f->var[i];
    if ((st = rd_name(f->fp, v->name)) || rd_u32(f->fp, &u)) { free_file(f); return st ? st : E_NOTNC; }
    if (u > ECNC_MAX_DIMS) { free_file(f); return E_MAXDIMS; }
    v->rank = (int)u;
    for (int k = 0; k < v->rank; ++k) {
      if (rd_u32(f->fp, &u) || (int)u >= f->ndim) { free_file(f); return E_NOTNC; }
      v->dimid[k] = (int)u;
    }
    v->is_rec = v->rank > 0 && v->dimid[0] == f->recdim;
    if ((st = rd_atts(f->fp, &v->natt, &v->att))) { free_file(f); return st; }
    if (rd_u32(f->fp, &u) || !tsize((int)u)) { free_file(f); return E_BADTYPE; }
    v->type = (int)u;
    if (rd_u32(f->fp, &u)) { free_file(f); return E_NOTNC; }
    v->vsize = u;
    if (f->version == 1) { if (rd_u32(f->fp, &u)) { free_file(f); return E_NOTNC; } v->begin = u; }
    else if (rd_u64(f->fp, &v->begin)) { free_file(f); return E_NOTNC; }
    /* (vsize is recomputed: the stored field saturates at 2^32-4 for large variables) */
    v->vsize = pad4(var_slab_elems(f, v) * tsize(v->type));
    if (v->is_rec) { f->recsize += v->vsize; ++nrec; }
  }
  if (nrec == 1)   /* a lone record variable is not padded */
    for (int i = 0; i < f->nvar; ++i)
      if (f->var[i].is_rec) f->recsize = var_slab_elems(f, &f->var[i]) * tsize(f->var[i].type);
  if (f->recdim >= 0) f->dim[f->recdim].len = f->numrecs;     /* report the current length */
  return 0;
}

/* ---- create / define ------------------------------------------------------------------------------- */
int ecnc_create(const char* path, int use_64bit_offset, int* ncid) {
  File* f;
  int st = new_slot(&f, ncid);
  if (st) return st;
  f->fp = fopen(path, "wb+");
  if (!f->fp) { st = errno ? errno : 13; f->used = 0; return st; }
  /* (use_64bit_offset: 0 = CDF-1, 1 = CDF-2, 2 = netCDF-4 / HDF5: ecnc_h5_enddef below) */
  f->writable = 1; f->define_mode = 1; f->version = use_64bit_offset == 2 ? 5 : use_64bit_offset ? 2 : 1;
  return 0;
}

int ecnc_def_dim(int ncid, const char* name, long long len, int* dimid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (!f->define_mode) return E_NOTINDEFINE;
  if (len <= 0) return E_UNLIMIT;      /* record dimensions are not written by this library */
  if (strlen(name) >= ECNC_MAX_NAME) return E_MAXNAME;
  for (int i = 0; i < f->ndim; ++i) if (!strcmp(f->dim[i].name, name)) return E_NAMEINUSE;
  f->dim = (Dim*)realloc(f->dim, (size_t)(f->ndim + 1) * sizeof(Dim));
  memset(&f->dim[f->ndim], 0, sizeof(Dim));
  strcpy(f->dim[f->ndim].name, name);
  f->dim[f->ndim].len = (uint64_t)len;
  *dimid = f->ndim++;
  return 0;
}

int ecnc_def_var(int ncid, const char* name, int xtype, int rank, const int* dimids, int* varid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (!f->define_mode) return E_NOTINDEFINE;
  if (!tsize(xtype)) return E_BADTYPE;
  if (rank < 0 || rank > ECNC_MAX_DIMS) return E_MAXDIMS;
  if (strlen(name) >= ECNC_MAX_NAME) return E_MAXNAME;
  for (int i = 0; i < f->nvar; ++i) if (!strcmp(f->var[i].name, name)) return E_NAMEINUSE;
  for (int k = 0; k < rank; ++k) if (dimids[k] < 0 || dimids[k] >= f->ndim) return E_BADDIM;
  f->var = (Var*)realloc(f->var, (size_t)(f->nvar + 1) * sizeof(Var));
  Var* v = &f->var[f->nvar];
  memset(v, 0, sizeof(Var));
  strcpy(v->name, name);
  v->rank = rank; v->type = xtype;
  for (int k = 0; k < rank; ++k) v->dimid[k] = dimids[k];
  *varid = f->nvar++;
  return 0;
}

static int att_slot(File* f, int varid, int** natt, Att*** att) {
  static Att* dummy;
  (void)dummy;
  if (varid == -1) { *natt = &f->ngatt; *att = &f->gatt; return 0; }
  if (varid < 0 || varid >= f->nvar) return E_NOTVAR;
  *natt = &f->var[varid].natt; *att = &f->var[varid].att;
  return 0;
}

/* mem -> file-type conversion of n values; memtype and xtype are T_* codes */
static int convert(const void* src, int stype, void* dst, int dtype, uint64_t n) {
  if ((stype == T_CHAR) != (dtype == T_CHAR)) return E_CHAR;
  if (stype == dtype) { memcpy(dst, src, n * tsize(stype)); return 0; }
  int range = 0;
  for (uint64_t i = 0; i < n; ++i) {
    double x;
    switch (stype) {
      case T_BYTE: x = ((const signed char*)src)[i]; break;
      case T_SHORT: x = ((const int16_t*)src)[i]; break;
      case T_INT: x = ((const int32_t*)src)[i]; break;
      case T_FLOAT: x = ((const float*)src)[i]; break;
      default: x = ((const double*)src)[i]; break;
    }
    switch (dtype) {
      case T_BYTE: if (x < -128 || x > 127) range = 1; ((signed char*)dst)[i] = (signed char)x; break;
      case T_SHORT: if (x < -32768 || x > 32767) range = 1; ((int16_t*)dst)[i] = (int16_t)x; break;
      case T_INT: if (x < -2147483648.0 || x > 2147483647.0) range = 1; ((int32_t*)dst)[i] = (int32_t)x; break;
      case T_FLOAT: ((float*)dst)[i] = (float)x; break;
      default: ((double*)dst)[i] = x; break;
    }
  }
  return range ? E_RANGE : 0;
}

int ecnc_put_att(int ncid, int varid, const char* name, int xtype, long long n, int memtype, const void* values) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (!f->define_mode) return E_NOTINDEFINE;
  if (!tsize(xtype) || !tsize(memtype) || n < 0) return E_BADTYPE;
  if (strlen(name) >= ECNC_MAX_NAME) return E_MAXNAME;
  int* natt; Att** att;
  int st = att_slot(f, varid, &natt, &att);
  if (st) return st;
  Att* a = NULL;
  for (int i = 0; i < *natt; ++i) if (!strcmp((*att)[i].name, name)) a = &(*att)[i];
  if (!a) {
    *att = (Att*)realloc(*att, (size_t)(*natt + 1) * sizeof(Att));
    a = &(*att)[(*natt)++];
    memset(a, 0, sizeof(Att));
    strcpy(a->name, name);
  }
  free(a->v);
  a->type = xtype; a->n = (uint64_t)n;
  a->v = calloc(1, (size_t)n * tsize(xtype) + 8);
  if (!a->v) return E_NOMEM;
  st = convert(values, memtype, a->v, xtype, (uint64_t)n);
  return st == E_RANGE ? 0 : st;
}

static const Att* find_att(File* f, int varid, const char* name, int* st) {
  int* natt; Att** att;
  *st = att_slot(f, varid, &natt, &att);
  if (*st) return NULL;
  for (int i = 0; i < *natt; ++i) if (!strcmp((*att)[i].name, name)) return &(*att)[i];
  *st = E_NOTATT;
  return NULL;
}

int ecnc_inq_att(int ncid, int varid, const char* name, int* xtype, long long* len) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  int st;
  const Att* a = find_att(f, varid, name, &st);
  if (!a) return st;
  if (xtype) *xtype = a->type;
  if (len) *len = (long long)a->n;
  return 0;
}

int ecnc_inq_attname(int ncid, int varid, int attnum, char* name, int name_cap) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  int* natt; Att** att;
  int st = att_slot(f, varid, &natt, &att);
  if (st) return st;
  if (attnum < 0 || attnum >= *natt) return E_NOTATT;
  snprintf(name, (size_t)name_cap, "%s", (*att)[attnum].name);
  return 0;
}

/* up to `cap` values, converted to memtype; *n = number of values of the attribute */
int ecnc_get_att(int ncid, int varid, const char* name, int memtype, void* out, long long cap, long long* n) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  int st;
  const Att* a = find_att(f, varid, name, &st);
  if (!a) return st;
  if (n) *n = (long long)a->n;
  const uint64_t m = a->n < (uint64_t)cap ? a->n : (uint64_t)cap;
  st = convert(a->v, a->type, out, memtype, m);
  return st;
}

int ecnc_copy_att(int ncid_in, int varid_in, const char* name, int ncid_out, int varid_out) {
  File* f = file_of(ncid_in);
  if (!f) return E_BADID;
  int st;
  const Att* a = find_att(f, varid_in, name, &st);
  if (!a) return st;
  return ecnc_put_att(ncid_out, varid_out, name, a->type, (long long)a->n, a->type, a->v);
}

int ecnc_enddef(int ncid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (!f->define_mode) return E_NOTINDEFINE;
  if (f->version == 5) return ecnc_h5_enddef(f);
  /* header size, then the variables one after the other */
  const uint64_t beg_bytes = f->version == 2 ? 8 : 4;
  uint64_t h = 8 + 8;
  for (int i = 0; i < f->ndim; ++i) h += name_bytes(f->dim[i].name) + 4;
  h += atts_bytes(f->ngatt, f->gatt) + 8;
  for (int v = 0; v < f->nvar; ++v) h += name_bytes(f->var[v].name) + 4 + 4 * (uint64_t)f->var[v].rank + atts_bytes(f->var[v].natt, f->var[v].att) + 8 + beg_bytes;
  uint64_t pos = pad4(h);
  for (int v = 0; v < f->nvar; ++v) {
    Var* x = &f->var[v];
    x->vsize = pad4(var_slab_elems(f, x) * tsize(x->type));
    x->begin = pos;
    pos += x->vsize;
  }
  if (f->version == 1 && pos > 0x7FFFFFFFull) {      /* does not fit 32-bit offsets: switch to CDF-2 and lay out again */
    f->version = 2;
    return ecnc_enddef(ncid);
  }
  rewind(f->fp);
  const unsigned char magic[4] = {'C', 'D', 'F', (unsigned char)f->version};
  fwrite(magic, 1, 4, f->fp);
  wr_u32(f->fp, 0);
  if (f->ndim) { wr_u32(f->fp, 0x0A); wr_u32(f->fp, (uint32_t)f->ndim); } else { wr_u32(f->fp, 0); wr_u32(f->fp, 0); }
  for (int i = 0; i < f->ndim; ++i) { wr_name(f->fp, f->dim[i].name); wr_u32(f->fp, (uint32_t)f->dim[i].len); }
  wr_atts(f->fp, f->ngatt, f->gatt);
  if (f->nvar) { wr_u32(f->fp, 0x0B); wr_u32(f->fp, (uint32_t)f->nvar); } else { wr_u32(f->fp, 0); wr_u32(f->fp, 0); }
  for (int v = 0; v < f->nvar; ++v) {
    const Var* x = &f->var[v];
    wr_name(f->fp, x->name);
    wr_u32(f->fp, (uint32_t)x->rank);
    for (int k = 0; k < x->rank; ++k) wr_u32(f->fp, (uint32_t)x->dimid[k]);
    wr_atts(f->fp, x->natt, x->att);
    wr_u32(f->fp, (uint32_t)x->type);
    wr_u32(f->fp, x->vsize > 0xFFFFFFFCull ? 0xFFFFFFFFu : (uint32_t)x->vsize);
    if (f->version == 2) wr_u64(f->fp, x->begin); else wr_u32(f->fp, (uint32_t)x->begin);
  }
  /* the data section exists from the start (zeros until written) */
  if (pos > 0) {
    if (fseeko(f->fp, (off_t)(pos - 1), SEEK_SET)) return errno;
    fputc(0, f->fp);
  }
  fflush(f->fp);
  f->define_mode = 0;
  return ferror(f->fp) ? 5 : 0;
}

int ecnc_close(int ncid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  int st = 0;
  if (f->writable && f->define_mode) st = ecnc_enddef(ncid);
  if (f->writable && fflush(f->fp)) st = errno;
  free_file(f);
  return st;
}

/* ---- inquiries ------------------------------------------------------------------------------------- */
int ecnc_inq(int ncid, int* ndims, int* nvars, int* ngatts, int* unlimdimid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (ndims) *ndims = f->ndim;
  if (nvars) *nvars = f->nvar;
  if (ngatts) *ngatts = f->ngatt;
  if (unlimdimid) *unlimdimid = f->recdim;
  return 0;
}
int ecnc_inq_dimid(int ncid, const char* name, int* dimid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  for (int i = 0; i < f->ndim; ++i) if (!strcmp(f->dim[i].name, name)) { *dimid = i; return 0; }
  return E_BADDIM;
}
int ecnc_inq_dim(int ncid, int dimid, char* name, int name_cap, long long* len) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (dimid < 0 || dimid >= f->ndim) return E_BADDIM;
  if (name) snprintf(name, (size_t)name_cap, "%s", f->dim[dimid].name);
  if (len) *len = (long long)f->dim[dimid].len;
  return 0;
}
int ecnc_inq_varid(int ncid, const char* name, int* varid) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  for (int i = 0; i < f->nvar; ++i) if (!strcmp(f->var[i].name, name)) { *varid = i; return 0; }
  return E_NOTVAR;
}
int ecnc_inq_var(int ncid, int varid, char* name, int name_cap, int* xtype, int* rank, int* dimids, int* natts) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (varid < 0 || varid >= f->nvar) return E_NOTVAR;
  const Var* v = &f->var[varid];
  if (name) snprintf(name, (size_t)name_cap, "%s", v->name);
  if (xtype) *xtype = v->type;
  if (rank) *rank = v->rank;
  if (dimids) for (int k = 0; k < v->rank; ++k) dimids[k] = v->dimid[k];
  if (natts) *natts = v->natt;
  return 0;
}

/* ---- data ------------------------------------------------------------------------------------------ */
/* hyperslab access: start/count in the file's (C, slowest first) dimension order; the memory buffer is contiguous in the
   same order.  nstart = 0 means the whole variable. */
static int vara(int ncid, int varid, int memtype, void* buf, int nidx, const long long* start, const long long* count, int writing) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (f->define_mode) return E_INDEFINE;
  if (writing && !f->writable) return E_PERM;
  if (varid < 0 || varid >= f->nvar) return E_NOTVAR;
  const Var* v = &f->var[varid];
  if ((memtype == T_CHAR) != (v->type == T_CHAR)) return E_CHAR;
  long long s[ECNC_MAX_DIMS + 1], c[ECNC_MAX_DIMS + 1], len[ECNC_MAX_DIMS + 1];
  const int r = v->rank;
  for (int k = 0; k < r; ++k) {
    len[k] = (long long)f->dim[v->dimid[k]].len;
    s[k] = nidx ? start[k] : 0;
    c[k] = nidx ? count[k] : len[k];
    if (s[k] < 0 || (c[k] > 0 && s[k] >= len[k] && len[k] > 0)) return E_INVALCOORDS;
    if (c[k] < 0 || s[k] + c[k] > len[k]) return E_EDGE;
  }
  if (nidx && nidx != r) return E_INVALCOORDS;
  if (f->version == 6) {
    if (writing) return E_PERM;
    for (int k = 0; k < r; ++k) if (c[k] == 0) return 0;
    return ecnc_h5_read(f, v, memtype, buf, s, c);
  }
  const size_t ts = tsize(v->type), ms = tsize(memtype);
  /* contiguous run: the last dimension (a scalar is one element) -- unless that is the record dimension itself (a
     one-dimensional record variable: one element per record) */
  const int nouter = (r == 1 && v->is_rec) ? 1 : (r ? r - 1 : 0);      /* dimensions iterated over */
  const long long run = (r && nouter == r - 1) ? c[r - 1] : 1;
  uint64_t nruns = 1;
  for (int k = 0; k < nouter; ++k) nruns *= (uint64_t)c[k];
  if (run == 0 || nruns == 0) return 0;
  /* element strides within one slab (the record dimension is handled through recsize) */
  uint64_t stride[ECNC_MAX_DIMS + 1];
  uint64_t acc = 1;
  for (int k = r - 1; k >= 0; --k) { stride[k] = acc; if (!(k == 0 && v->is_rec)) acc *= (uint64_t)len[k]; }
  unsigned char* tmp = (unsigned char*)malloc((size_t)run * ts + 8);
  if (!tmp) return E_NOMEM;
  long long idx[ECNC_MAX_DIMS + 1];
  for (int k = 0; k < r; ++k) idx[k] = 0;
  int st = 0, range = 0;
  unsigned char* mem = (unsigned char*)buf;
  for (uint64_t it = 0; it < nruns && !st; ++it) {
    uint64_t off = v->begin;
    for (int k = 0; k < r; ++k) {
      const uint64_t i = (uint64_t)(s[k] + (k >= nouter ? 0 : idx[k]));
      if (k == 0 && v->is_rec) off += i * f->recsize; else off += i * stride[k] * ts;
    }
    if (fseeko(f->fp, (off_t)off, SEEK_SET)) { st = errno; break; }
    if (writing) {
      int cs = convert(mem, memtype, tmp, v->type, (uint64_t)run);
      if (cs == E_RANGE) range = 1; else if (cs) { st = cs; break; }
      if (f->version != 5) swap_in_place(tmp, ts, (uint64_t)run);      /* (an HDF5 file of this library holds little-endian values) */
      if (fwrite(tmp, ts, (size_t)run, f->fp) != (size_t)run) { st = errno ? errno : 5; break; }
    } else {
      if (fread(tmp, ts, (size_t)run, f->fp) != (size_t)run) { st = E_EDGE; break; }
      if (f->version != 5) swap_in_place(tmp, ts, (uint64_t)run);
      int cs = convert(tmp, v->type, mem, memtype, (uint64_t)run);
      if (cs == E_RANGE) range = 1; else if (cs) { st = cs; break; }
    }
    mem += (size_t)run * ms;
    for (int k = nouter - 1; k >= 0; --k) { if (++idx[k] < c[k]) break; idx[k] = 0; }
  }
  free(tmp);
  return st ? st : (range ? E_RANGE : 0);
}

int ecnc_get_vara(int ncid, int varid, int memtype, void* buf, int nidx, const long long* start, const long long* count) {
  return vara(ncid, varid, memtype, buf, nidx, start, count, 0);
}
int ecnc_put_vara(int ncid, int varid, int memtype, const void* buf, int nidx, const long long* start, const long long* count) {
  return vara(ncid, varid, memtype, (void*)buf, nidx, start, count, 1);
}

/* number of elements of a whole variable (records included) */
int ecnc_var_elems(int ncid, int varid, long long* n) {
  File* f = file_of(ncid);
  if (!f) return E_BADID;
  if (varid < 0 || varid >= f->nvar) return E_NOTVAR;
  uint64_t m = 1;
  for (int k = 0; k < f->var[varid].rank; ++k) m *= f->dim[f->var[varid].dimid[k]].len;
  *n = (long long)m;
  return 0;
}

const char* ecnc_strerror(int st) {
  switch (st) {
    case 0: return "No error";
    case E_BADID: return "NetCDF: Not a valid ID";
    case E_NFILE: return "NetCDF: Too many files open";
    case E_INVAL: return "NetCDF: Invalid argument";
    case E_PERM: return "NetCDF: Write to read only";
    case E_NOTINDEFINE: return "NetCDF: Operation not allowed in data mode";
    case E_INDEFINE: return "NetCDF: Operation not allowed in define mode";
    case E_INVALCOORDS: return "NetCDF: Index exceeds dimension bound";
    case E_MAXDIMS: return "NetCDF: NC_MAX_DIMS exceeded";
    case E_NAMEINUSE: return "NetCDF: String match to name in use";
    case E_NOTATT: return "NetCDF: Attribute not found";
    case E_BADTYPE: return "NetCDF: Not a valid data type or _FillValue type mismatch";
    case E_BADDIM: return "NetCDF: Invalid dimension ID or name";
    case E_NOTVAR: return "NetCDF: Variable not found";
    case E_NOTNC: return "NetCDF: Unknown file format (this library reads classic CDF-1 / CDF-2 files)";
    case E_MAXNAME: return "NetCDF: NC_MAX_NAME exceeded";
    case E_UNLIMIT: return "NetCDF: NC_UNLIMITED size already in use (this library writes fixed dimensions only)";
    case E_CHAR: return "NetCDF: Attempt to convert between text & numbers";
    case E_EDGE: return "NetCDF: Start+count exceeds dimension bound";
    case E_RANGE: return "NetCDF: Numeric conversion not representable";
    case E_NOMEM: return "NetCDF: Memory allocation (malloc) failure";
    case E_NOHDF5: return "NetCDF: the file is netCDF-4 / HDF5 and no HDF5 library could be loaded (libhdf5.so; set ECRAD_HDF5_LIB)";
    case E_HDFERR: return "NetCDF: HDF error";
    default: return st > 0 ? strerror(st) : "NetCDF: Unknown error";
  }
}


/* ==== netCDF-4 / HDF5 output ========================================================================================
 * nf90_create(..., NF90_HDF5) -- easy_netcdf's is_hdf5_file (utilities/easy_netcdf.F90:212-245), the driver namelist's
 * do_write_hdf5 -- gives a file in the HDF5 format, written directly (the image has no libnetcdf / libhdf5 for the Fortran
 * host): the same subset, laid out the same way, as the Python host's writer (ecrad_amd/hdf5file.py: write_nc4, which
 * tests/test_hdf5_output.py reads back with the HDF5 library's own C API) --
 *   superblock version 0; the root group in the original symbol-table form (one B-tree node, a local heap, one symbol node);
 *   version-1 object headers; contiguous little-endian datasets of int32 / float32 / float64; attributes in the object
 *   headers; one global heap collection for the variable-length DIMENSION_LIST attributes;
 * plus what makes an HDF5 file a netCDF-4 file (libnetcdf's nc4hdf.c): every dimension a "dimension scale" dataset (CLASS,
 * NAME, _Netcdf4Dimid, REFERENCE_LIST), every variable with DIMENSION_LIST and _Netcdf4Coordinates, _NCProperties on the root.
 * Define mode collects dimensions, variables and attributes as for the classic format; ecnc_h5_enddef lays the whole file out
 * (every size is known then), writes the metadata and reserves the data; ecnc_put_vara then writes into the datasets at
 * Var.begin exactly as it does for a classic file, minus the byte swap.  Fixed dimensions; numeric variables (byte, short, int,
 * float, double: radiation_save.F90 writes float / double / int); files written this way are not read back by this library (ecnc_open reads classic).
 */
typedef struct { unsigned char* p; size_t n, cap; } HBuf;
static void hb_need(HBuf* b, size_t more) {
  if (b->n + more <= b->cap) return;
  size_t cap = b->cap ? b->cap : 256;
  while (cap < b->n + more) cap *= 2;
  b->p = (unsigned char*)realloc(b->p, cap);
  b->cap = cap;
}
static void hb_put(HBuf* b, const void* src, size_t n) { hb_need(b, n); if (n) memcpy(b->p + b->n, src, n); b->n += n; }
static void hb_zero(HBuf* b, size_t n) { hb_need(b, n); memset(b->p + b->n, 0, n); b->n += n; }
static void hb_u8(HBuf* b, unsigned v) { unsigned char c = (unsigned char)v; hb_put(b, &c, 1); }
static void hb_u16(HBuf* b, unsigned v) { unsigned char c[2] = {(unsigned char)v, (unsigned char)(v >> 8)}; hb_put(b, c, 2); }
static void hb_u32(HBuf* b, uint32_t v) { unsigned char c[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; hb_put(b, c, 4); }
static void hb_u64(HBuf* b, uint64_t v) { hb_u32(b, (uint32_t)v); hb_u32(b, (uint32_t)(v >> 32)); }
static void hb_pad8(HBuf* b, size_t from) { const size_t len = b->n - from; hb_zero(b, (8 - len % 8) % 8); }
static void hb_free(HBuf* b) { free(b->p); b->p = NULL; b->n = b->cap = 0; }
#define H5_UNDEF 0xFFFFFFFFFFFFFFFFull
#define H5_LEAF_K 64
#define H5_INTERNAL_K 16

/* ---- datatype messages */
static void h5_dt_float(HBuf* b, int nbytes) {
  if (nbytes == 8) { hb_u8(b, 0x11); hb_u8(b, 0x20); hb_u8(b, 0x3F); hb_u8(b, 0); hb_u32(b, 8); hb_u16(b, 0); hb_u16(b, 64); hb_u8(b, 52); hb_u8(b, 11); hb_u8(b, 0); hb_u8(b, 52); hb_u32(b, 1023); }
  else { hb_u8(b, 0x11); hb_u8(b, 0x20); hb_u8(b, 0x1F); hb_u8(b, 0); hb_u32(b, 4); hb_u16(b, 0); hb_u16(b, 32); hb_u8(b, 23); hb_u8(b, 8); hb_u8(b, 0); hb_u8(b, 23); hb_u32(b, 127); }
}
static void h5_dt_int(HBuf* b, unsigned nbytes) { hb_u8(b, 0x10); hb_u8(b, 0x08); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, nbytes); hb_u16(b, 0); hb_u16(b, 8 * nbytes); }      /* signed, little-endian */
static void h5_dt_int32(HBuf* b) { h5_dt_int(b, 4); }
static void h5_dt_string(HBuf* b, uint32_t n) { hb_u8(b, 0x13); hb_u8(b, 0); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, n); }
static void h5_dt_objref(HBuf* b) { hb_u8(b, 0x17); hb_u8(b, 0); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, 8); }
static void h5_dt_vlen_objref(HBuf* b) { hb_u8(b, 0x19); hb_u8(b, 0); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, 16); h5_dt_objref(b); }
static void h5_dt_member(HBuf* b, const char* name, uint32_t offset, int is_ref) {
  const size_t from = b->n;
  hb_put(b, name, strlen(name) + 1); hb_pad8(b, from);
  hb_u32(b, offset); hb_u8(b, 0); hb_zero(b, 3); hb_u32(b, 0); hb_u32(b, 0); for (int k = 0; k < 4; ++k) hb_u32(b, 0);
  if (is_ref) h5_dt_objref(b); else h5_dt_int32(b);
}
static void h5_dt_reference_list(HBuf* b) {      /* compound { object reference "dataset" @0; int32 "dimension" @8 }, 16 bytes */
  hb_u8(b, 0x16); hb_u8(b, 2); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, 16);
  h5_dt_member(b, "dataset", 0, 1); h5_dt_member(b, "dimension", 8, 0);
}
static void h5_dataspace(HBuf* b, int rank, const uint64_t* shape) {
  hb_u8(b, 1); hb_u8(b, (unsigned)rank); hb_u8(b, 0); hb_u8(b, 0); hb_u32(b, 0);
  for (int k = 0; k < rank; ++k) hb_u64(b, shape[k]);
}
/* attribute message: name, datatype and dataspace each padded to 8 bytes, then the data */
static void h5_attribute(HBuf* out, const char* name, const HBuf* dtype, int rank, const uint64_t* shape, const void* data, size_t nbytes) {
  HBuf ds = {0};
  h5_dataspace(&ds, rank, shape);
  const size_t nm = strlen(name) + 1;
  hb_u8(out, 1); hb_u8(out, 0); hb_u16(out, (unsigned)nm); hb_u16(out, (unsigned)dtype->n); hb_u16(out, (unsigned)ds.n);
  size_t from = out->n; hb_put(out, name, nm); hb_pad8(out, from);
  from = out->n; hb_put(out, dtype->p, dtype->n); hb_pad8(out, from);
  from = out->n; hb_put(out, ds.p, ds.n); hb_pad8(out, from);
  hb_put(out, data, nbytes);
  hb_free(&ds);
}
static void h5_attr_string(HBuf* out, const char* name, const char* value, size_t len /* without the terminator */) {
  HBuf dt = {0};
  char* v = (char*)calloc(1, len + 1);
  memcpy(v, value, len);
  h5_dt_string(&dt, (uint32_t)(len + 1));
  h5_attribute(out, name, &dt, 0, NULL, v, len + 1);
  free(v);
  hb_free(&dt);
}
/* a numeric attribute as the Python writer stores it: integers (byte, short, int) as int32, float as float32, double as float64 */
static void h5_attr_numeric(HBuf* out, const Att* a) {
  HBuf dt = {0}, data = {0};
  const uint64_t shape[1] = {a->n};
  if (a->type == T_FLOAT) { h5_dt_float(&dt, 4); hb_put(&data, a->v, (size_t)a->n * 4); }
  else if (a->type == T_DOUBLE) { h5_dt_float(&dt, 8); hb_put(&data, a->v, (size_t)a->n * 8); }
  else {
    h5_dt_int32(&dt);
    for (uint64_t i = 0; i < a->n; ++i)
      hb_u32(&data, (uint32_t)(a->type == T_BYTE ? ((const signed char*)a->v)[i] : a->type == T_SHORT ? ((const int16_t*)a->v)[i] : ((const int32_t*)a->v)[i]));
  }
  h5_attribute(out, a->name, &dt, 1, shape, data.p, data.n);
  hb_free(&dt); hb_free(&data);
}
static void h5_user_attr(HBuf* out, const Att* a) {
  if (a->type == T_CHAR) h5_attr_string(out, a->name, (const char*)a->v, (size_t)a->n);
  else h5_attr_numeric(out, a);
}

typedef struct {
  char name[ECNC_MAX_NAME];
  int varid;            /* the variable this dataset is, or -1: a dimension without a variable of its name */
  int dimid;            /* the dimension this dataset is the scale of, or -1 */
  int is_coord;         /* a scale that is also a (coordinate) variable */
  int rank, type;
  int dn[ECNC_MAX_DIMS];        /* dimension ids of a variable (none for a pure scale) */
  int ndn;
  uint64_t shape[ECNC_MAX_DIMS];
  uint64_t addr, data_addr, nbytes;
} H5Obj;
typedef struct { int obj, k, dim; } H5Ref;      /* (dataset, index of the dimension in it, dimension) */

static int h5_cmp_obj(const void* a, const void* b) { return strcmp(((const H5Obj*)a)->name, ((const H5Obj*)b)->name); }

/* messages of one object header (version 1): type, size of the padded body, flags, 3 reserved bytes, body */
static void h5_message(HBuf* hdr, unsigned type, const HBuf* body, int* nmsg) {
  const size_t padded = (body->n + 7) & ~(size_t)7;
  hb_u16(hdr, type); hb_u16(hdr, (unsigned)padded); hb_u8(hdr, 0); hb_zero(hdr, 3);
  hb_put(hdr, body->p, body->n); hb_zero(hdr, padded - body->n);
  ++*nmsg;
}
static void h5_object_header(HBuf* out, const HBuf* messages, int nmsg) {
  hb_u8(out, 1); hb_u8(out, 0); hb_u16(out, (unsigned)nmsg); hb_u32(out, 1); hb_u32(out, (uint32_t)messages->n); hb_zero(out, 4);
  hb_put(out, messages->p, messages->n);
}

static int ecnc_h5_enddef(File* f) {
  for (int d = 0; d < f->ndim; ++d) if (f->dim[d].len == 0) return E_UNLIMIT;
  for (int v = 0; v < f->nvar; ++v) if (f->var[v].type == T_CHAR) return E_BADTYPE;      /* (no text variables: radiation_save.F90 writes none) */
  const int nobj_max = f->nvar + f->ndim;
  if (nobj_max > 2 * H5_LEAF_K) return E_MAXDIMS;      /* (one symbol-table node: up to 128 dimensions + variables) */
  H5Obj* objs = (H5Obj*)calloc((size_t)nobj_max + 1, sizeof(H5Obj));
  if (!objs) return E_NOMEM;
  int nobj = 0;
  for (int v = 0; v < f->nvar; ++v) {
    const Var* x = &f->var[v];
    H5Obj* o = &objs[nobj++];
    strcpy(o->name, x->name);
    o->varid = v; o->dimid = -1; o->rank = x->rank; o->type = x->type; o->ndn = x->rank;
    o->nbytes = tsize(x->type);
    for (int k = 0; k < x->rank; ++k) { o->dn[k] = x->dimid[k]; o->shape[k] = f->dim[x->dimid[k]].len; o->nbytes *= o->shape[k]; }
  }
  /* a dimension without a variable of its name becomes a float32 dataset that is never written (netCDF-4's own habit) */
  for (int d = 0; d < f->ndim; ++d) {
    int found = -1;
    for (int i = 0; i < nobj; ++i) if (objs[i].varid >= 0 && !strcmp(objs[i].name, f->dim[d].name)) found = i;
    if (found >= 0) {
      if (!(objs[found].ndn == 1 && objs[found].dn[0] == d)) { free(objs); return E_NAMEINUSE; }      /* named like a dimension, not its coordinate variable */
      objs[found].dimid = d; objs[found].is_coord = 1;
    } else {
      H5Obj* o = &objs[nobj++];
      strcpy(o->name, f->dim[d].name);
      o->varid = -1; o->dimid = d; o->rank = 1; o->type = T_FLOAT; o->ndn = 0; o->shape[0] = f->dim[d].len; o->nbytes = 4 * o->shape[0];
    }
  }
  qsort(objs, (size_t)nobj, sizeof(H5Obj), h5_cmp_obj);      /* symbol-table entries are kept in strcmp order */
  int* scale_of_dim = (int*)calloc((size_t)f->ndim + 1, sizeof(int));
  for (int i = 0; i < nobj; ++i) if (objs[i].dimid >= 0) scale_of_dim[objs[i].dimid] = i;
  /* global heap: one object (a sequence of one object reference) per (variable, dimension it is attached to) */
  H5Ref* gitems = (H5Ref*)calloc((size_t)nobj * ECNC_MAX_DIMS + 1, sizeof(H5Ref));
  int ngitems = 0;
  for (int i = 0; i < nobj; ++i)
    if (objs[i].varid >= 0)
      for (int k = 0; k < objs[i].ndn; ++k) {
        if (scale_of_dim[objs[i].dn[k]] == i) continue;      /* a coordinate variable is not attached to itself */
        gitems[ngitems].obj = i; gitems[ngitems].k = k; gitems[ngitems].dim = objs[i].dn[k]; ++ngitems;
      }
  const uint64_t gheap_used = 16 + 24 * (uint64_t)ngitems;
  uint64_t gheap_size = ((gheap_used + 16 + 4095) / 4096) * 4096;
  if (gheap_size < 4096) gheap_size = 4096;

  /* local heap of the root group: the names */
  HBuf heap = {0};
  hb_zero(&heap, 8);
  uint64_t* name_off = (uint64_t*)calloc((size_t)nobj + 1, sizeof(uint64_t));
  for (int i = 0; i < nobj; ++i) { name_off[i] = heap.n; const size_t from = heap.n; hb_put(&heap, objs[i].name, strlen(objs[i].name) + 1); hb_pad8(&heap, from); }
  const uint64_t heap_free = heap.n;
  hb_u64(&heap, 1); hb_u64(&heap, 32); hb_zero(&heap, 16);
  const uint64_t btree_size = 24 + (2 * H5_INTERNAL_K + 1) * 8 + 2 * H5_INTERNAL_K * 8;
  const uint64_t snod_size = 8 + 2 * H5_LEAF_K * 40;

  uint64_t gheap_addr = 0, btree_addr = 0, heap_addr = 0;
  HBuf root = {0};
  HBuf* hdr = (HBuf*)calloc((size_t)nobj + 1, sizeof(HBuf));
  uint64_t root_addr = 96, heap_data_addr = 0, snod_addr = 0, eof = 0;
  /* two passes: the sizes of the headers do not depend on the addresses inside them; the second pass has the addresses */
  for (int pass = 0; pass < 2; ++pass) {
    /* root group: symbol-table message + _NCProperties + the global attributes */
    {
      HBuf msgs = {0}, body = {0};
      int nmsg = 0;
      hb_u64(&body, btree_addr); hb_u64(&body, heap_addr);
      h5_message(&msgs, 0x0011, &body, &nmsg);
      body.n = 0;
      h5_attr_string(&body, "_NCProperties", "version=2,ecrad_amd=1", strlen("version=2,ecrad_amd=1"));
      h5_message(&msgs, 0x000C, &body, &nmsg);
      for (int a = 0; a < f->ngatt; ++a) { body.n = 0; h5_user_attr(&body, &f->gatt[a]); h5_message(&msgs, 0x000C, &body, &nmsg); }
      root.n = 0;
      h5_object_header(&root, &msgs, nmsg);
      hb_free(&msgs); hb_free(&body);
    }
    for (int i = 0; i < nobj; ++i) {
      const H5Obj* o = &objs[i];
      HBuf msgs = {0}, body = {0};
      int nmsg = 0;
      h5_dataspace(&body, o->rank, o->shape); h5_message(&msgs, 0x0001, &body, &nmsg);
      body.n = 0;
      if (o->type == T_FLOAT || o->type == T_DOUBLE) h5_dt_float(&body, o->type == T_DOUBLE ? 8 : 4); else h5_dt_int(&body, (unsigned)tsize(o->type));
      h5_message(&msgs, 0x0003, &body, &nmsg);
      body.n = 0; hb_u8(&body, 2); hb_u8(&body, 2); hb_u8(&body, 2); hb_u8(&body, 1); hb_u32(&body, 0); h5_message(&msgs, 0x0005, &body, &nmsg);
      body.n = 0; hb_u8(&body, 3); hb_u8(&body, 1); hb_u64(&body, o->data_addr); hb_u64(&body, o->nbytes); h5_message(&msgs, 0x0008, &body, &nmsg);
      if (o->dimid >= 0) {      /* a dimension scale */
        body.n = 0; h5_attr_string(&body, "CLASS", "DIMENSION_SCALE", 15); h5_message(&msgs, 0x000C, &body, &nmsg);
        body.n = 0;
        if (o->is_coord) h5_attr_string(&body, "NAME", o->name, strlen(o->name));
        else {
          char nm[128];
          snprintf(nm, sizeof nm, "This is a netCDF dimension but not a netCDF variable.%10llu", (unsigned long long)o->shape[0]);
          h5_attr_string(&body, "NAME", nm, strlen(nm));
        }
        h5_message(&msgs, 0x000C, &body, &nmsg);
        {
          HBuf dt = {0}; h5_dt_int32(&dt);
          const int32_t id = o->dimid;
          body.n = 0; h5_attribute(&body, "_Netcdf4Dimid", &dt, 0, NULL, &id, 4); h5_message(&msgs, 0x000C, &body, &nmsg);
          hb_free(&dt);
        }
        /* REFERENCE_LIST: the (dataset, index) pairs that use this dimension, in the order of the datasets */
        HBuf refs = {0};
        uint64_t nrefs = 0;
        for (int v = 0; v < f->nvar; ++v) {      /* (in the order the variables were defined, as the Python writer's `users`) */
          int u = -1;
          for (int j = 0; j < nobj; ++j) if (objs[j].varid == v) u = j;
          if (u == i) continue;
          for (int k = 0; k < objs[u].ndn; ++k)
            if (objs[u].dn[k] == o->dimid) { hb_u64(&refs, objs[u].addr); hb_u32(&refs, (uint32_t)k); hb_zero(&refs, 4); ++nrefs; }
        }
        if (nrefs) {
          HBuf dt = {0}; h5_dt_reference_list(&dt);
          body.n = 0; h5_attribute(&body, "REFERENCE_LIST", &dt, 1, &nrefs, refs.p, refs.n); h5_message(&msgs, 0x000C, &body, &nmsg);
          hb_free(&dt);
        }
        hb_free(&refs);
      }
      int attached = 0;
      for (int k = 0; k < o->ndn; ++k) if (scale_of_dim[o->dn[k]] != i) attached = 1;
      if (attached) {
        HBuf dt = {0}, data = {0};
        h5_dt_vlen_objref(&dt);
        for (int k = 0; k < o->ndn; ++k) {
          uint32_t index = 0;
          for (int q = 0; q < ngitems; ++q) if (gitems[q].obj == i && gitems[q].k == k) index = (uint32_t)q + 1;
          hb_u32(&data, 1); hb_u64(&data, gheap_addr); hb_u32(&data, index);
        }
        const uint64_t n = (uint64_t)o->ndn;
        body.n = 0; h5_attribute(&body, "DIMENSION_LIST", &dt, 1, &n, data.p, data.n); h5_message(&msgs, 0x000C, &body, &nmsg);
        hb_free(&dt); hb_free(&data);
      }
      if (o->ndn > 0 && !(o->dimid >= 0 && !o->is_coord)) {
        HBuf dt = {0}, data = {0};
        h5_dt_int32(&dt);
        for (int k = 0; k < o->ndn; ++k) hb_u32(&data, (uint32_t)o->dn[k]);
        const uint64_t n = (uint64_t)o->ndn;
        body.n = 0; h5_attribute(&body, "_Netcdf4Coordinates", &dt, 1, &n, data.p, data.n); h5_message(&msgs, 0x000C, &body, &nmsg);
        hb_free(&dt); hb_free(&data);
      }
      if (o->varid >= 0)
        for (int a = 0; a < f->var[o->varid].natt; ++a) { body.n = 0; h5_user_attr(&body, &f->var[o->varid].att[a]); h5_message(&msgs, 0x000C, &body, &nmsg); }
      hdr[i].n = 0;
      h5_object_header(&hdr[i], &msgs, nmsg);
      hb_free(&msgs); hb_free(&body);
    }
    if (pass == 0) {      /* the layout */
      uint64_t pos = root_addr + root.n;
      btree_addr = pos; pos += btree_size;
      heap_addr = pos; pos += 32;
      heap_data_addr = pos; pos += heap.n;
      snod_addr = pos; pos += snod_size;
      gheap_addr = pos; pos += gheap_size;
      for (int i = 0; i < nobj; ++i) { objs[i].addr = pos; pos += hdr[i].n; }
      for (int i = 0; i < nobj; ++i) {
        objs[i].data_addr = H5_UNDEF;
        if (objs[i].varid >= 0 && objs[i].nbytes > 0) { pos += (8 - pos % 8) % 8; objs[i].data_addr = pos; pos += objs[i].nbytes; }
      }
      eof = pos;
    }
  }
  /* ---- write */
  HBuf sb = {0};
  static const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  hb_put(&sb, sig, 8);
  hb_u8(&sb, 0); hb_u8(&sb, 0); hb_u8(&sb, 0); hb_u8(&sb, 0); hb_u8(&sb, 0); hb_u8(&sb, 8); hb_u8(&sb, 8); hb_u8(&sb, 0);
  hb_u16(&sb, H5_LEAF_K); hb_u16(&sb, H5_INTERNAL_K); hb_u32(&sb, 0);
  hb_u64(&sb, 0); hb_u64(&sb, H5_UNDEF); hb_u64(&sb, eof); hb_u64(&sb, H5_UNDEF);
  hb_u64(&sb, 0); hb_u64(&sb, root_addr); hb_u32(&sb, 1); hb_u32(&sb, 0); hb_u64(&sb, btree_addr); hb_u64(&sb, heap_addr);
  int st = 0;
  FILE* fp = f->fp;
  rewind(fp);
  fwrite(sb.p, 1, sb.n, fp);                                  /* 96 bytes */
  fwrite(root.p, 1, root.n, fp);
  {
    HBuf bt = {0};
    hb_put(&bt, "TREE", 4); hb_u8(&bt, 0); hb_u8(&bt, 0); hb_u16(&bt, 1); hb_u64(&bt, H5_UNDEF); hb_u64(&bt, H5_UNDEF);
    hb_u64(&bt, 0); hb_u64(&bt, snod_addr); hb_u64(&bt, nobj ? name_off[nobj - 1] : 0);
    hb_zero(&bt, (size_t)btree_size - bt.n);
    fwrite(bt.p, 1, bt.n, fp);
    hb_free(&bt);
  }
  {
    HBuf hh = {0};
    hb_put(&hh, "HEAP", 4); hb_u8(&hh, 0); hb_zero(&hh, 3); hb_u64(&hh, heap.n); hb_u64(&hh, heap_free); hb_u64(&hh, heap_data_addr);
    fwrite(hh.p, 1, hh.n, fp);
    hb_free(&hh);
  }
  fwrite(heap.p, 1, heap.n, fp);
  {
    HBuf sn = {0};
    hb_put(&sn, "SNOD", 4); hb_u8(&sn, 1); hb_u8(&sn, 0); hb_u16(&sn, (unsigned)nobj);
    for (int i = 0; i < nobj; ++i) { hb_u64(&sn, name_off[i]); hb_u64(&sn, objs[i].addr); hb_u32(&sn, 0); hb_u32(&sn, 0); hb_zero(&sn, 16); }
    hb_zero(&sn, (size_t)snod_size - sn.n);
    fwrite(sn.p, 1, sn.n, fp);
    hb_free(&sn);
  }
  {
    HBuf gh = {0};
    hb_put(&gh, "GCOL", 4); hb_u8(&gh, 1); hb_zero(&gh, 3); hb_u64(&gh, gheap_size);
    for (int q = 0; q < ngitems; ++q) { hb_u16(&gh, (unsigned)q + 1); hb_u16(&gh, 1); hb_zero(&gh, 4); hb_u64(&gh, 8); hb_u64(&gh, objs[scale_of_dim[gitems[q].dim]].addr); }
    { const uint64_t left = gheap_size - gh.n; hb_u16(&gh, 0); hb_u16(&gh, 0); hb_zero(&gh, 4); hb_u64(&gh, left); }      /* object 0: the free space (size includes this header) */
    hb_zero(&gh, (size_t)gheap_size - gh.n);
    fwrite(gh.p, 1, gh.n, fp);
    hb_free(&gh);
  }
  for (int i = 0; i < nobj; ++i) fwrite(hdr[i].p, 1, hdr[i].n, fp);
  /* the data section exists from the start (zeros until written) */
  if (eof > 0) {
    if (fseeko(fp, (off_t)(eof - 1), SEEK_SET)) st = errno; else fputc(0, fp);
  }
  fflush(fp);
  for (int i = 0; i < nobj; ++i)
    if (objs[i].varid >= 0) { f->var[objs[i].varid].begin = objs[i].data_addr; f->var[objs[i].varid].vsize = objs[i].nbytes; }
  f->define_mode = 0;
  if (ferror(fp) && !st) st = 5;
  for (int i = 0; i < nobj; ++i) hb_free(&hdr[i]);
  hb_free(&root); hb_free(&heap); hb_free(&sb);
  free(hdr); free(objs); free(gitems); free(name_off); free(scale_of_dim);
  return st;
}

/* ---- the four Fortran-77 entry points utilities/easy_netcdf.F90:3262-3324 calls as externals ----------- */
/* (gfortran / flang name mangling: lower case + underscore; arguments by reference; varid is 1-based) */
int nf_get_var_double_(const int* ncid, const int* varid, double* vals) { return vara(*ncid, *varid - 1, T_DOUBLE, vals, 0, NULL, NULL, 0); }
int nf_put_var_double_(const int* ncid, const int* varid, const double* vals) { return vara(*ncid, *varid - 1, T_DOUBLE, (void*)vals, 0, NULL, NULL, 1); }
int nf_get_var_int_(const int* ncid, const int* varid, int* vals) { return vara(*ncid, *varid - 1, T_INT, vals, 0, NULL, NULL, 0); }
int nf_put_var_int_(const int* ncid, const int* varid, const int* vals) { return vara(*ncid, *varid - 1, T_INT, (void*)vals, 0, NULL, NULL, 1); }

/* ==================================================================================================================
 * netCDF-4 INPUT: files in HDF5 format are read through the HDF5 library, loaded at run time (dlopen) -- the host needs it only
 * when it is handed such a file, and says so when it is not there (E_NOHDF5).  What libnetcdf would do on top of libhdf5 for the
 * subset a radiation driver meets (utilities/easy_netcdf.F90:133-200 opens whatever the library opens):
 *   dimensions  = datasets of the root group with CLASS = "DIMENSION_SCALE"; their id is `_Netcdf4Dimid` where the file has it,
 *                 the order of appearance otherwise; a dataset whose NAME begins "This is a netCDF dimension but not a netCDF
 *                 variable" is a dimension only, every other dataset is a variable (a coordinate variable is both);
 *   a variable's dimensions = the scales its DIMENSION_LIST attribute refers to (object references, dereferenced and looked up
 *                 by name); a dataset without the attribute (a plain HDF5 file) gets anonymous dimensions by length;
 *   types       = 1/2/4-byte integers, 8-byte integers (read as double), float, double, fixed-length strings of one byte (char);
 *   attributes  = numeric arrays and fixed- or variable-length strings; HDF5's and netCDF-4's own book-keeping attributes are hidden.
 * Groups, user-defined types and NC_STRING variables are not read (the reference's files have none).  Values are converted by the HDF5
 * library to the caller's memory type; hyperslabs map one to one.
 * ================================================================================================================== */
#include <dlfcn.h>
typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;
typedef struct { size_t len; void* p; } hvl_t;
typedef herr_t (*H5A_operator2_t)(hid_t loc, const char* name, const void* ainfo, void* op_data);
static struct {
  void* lib;
  int tried;
  herr_t (*open)(void);
  hid_t (*Fopen)(const char*, unsigned, hid_t);
  herr_t (*Fclose)(hid_t);
  herr_t (*Gget_num_objs)(hid_t, hsize_t*);
  long (*Gget_objname_by_idx)(hid_t, hsize_t, char*, size_t);
  int (*Gget_objtype_by_idx)(hid_t, hsize_t);
  hid_t (*Dopen2)(hid_t, const char*, hid_t);
  herr_t (*Dclose)(hid_t);
  hid_t (*Dget_space)(hid_t);
  hid_t (*Dget_type)(hid_t);
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
  herr_t (*Dvlen_reclaim)(hid_t, hid_t, hid_t, void*);
  int (*Sget_simple_extent_ndims)(hid_t);
  int (*Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
  long long (*Sget_simple_extent_npoints)(hid_t);
  hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*);
  herr_t (*Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*);
  herr_t (*Sclose)(hid_t);
  int (*Tget_class)(hid_t);
  size_t (*Tget_size)(hid_t);
  int (*Tis_variable_str)(hid_t);
  hid_t (*Tcopy)(hid_t);
  herr_t (*Tset_size)(hid_t, size_t);
  herr_t (*Tclose)(hid_t);
  int (*Aexists)(hid_t, const char*);
  hid_t (*Aopen)(hid_t, const char*, hid_t);
  herr_t (*Aclose)(hid_t);
  hid_t (*Aget_type)(hid_t);
  hid_t (*Aget_space)(hid_t);
  herr_t (*Aread)(hid_t, hid_t, void*);
  herr_t (*Aiterate2)(hid_t, int, int, hsize_t*, H5A_operator2_t, void*);
  hid_t (*Rdereference2)(hid_t, hid_t, int, const void*);
  hid_t (*Rdereference1)(hid_t, int, const void*);
  long (*Iget_name)(hid_t, char*, size_t);
  herr_t (*Oclose)(hid_t);
  herr_t (*Eset_auto2)(hid_t, void*, void*);
  herr_t (*free_memory)(void*);
  hid_t t_double, t_float, t_int, t_short, t_schar, t_c_s1, t_ref_obj;
} H5;

static int h5_load(void) {
  if (H5.tried) return H5.lib ? 0 : E_NOHDF5;
  H5.tried = 1;
  const char* names[] = {getenv("ECRAD_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310",
                         "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"};
  void* lib = NULL;
  for (size_t i = 0; i < sizeof names / sizeof names[0] && !lib; ++i)
    if (names[i] && names[i][0]) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return E_NOHDF5;
  int missing = 0;
#define H5SYM(field, sym) do { *(void**)&H5.field = dlsym(lib, sym); if (!H5.field) missing = 1; } while (0)
  H5SYM(open, "H5open"); H5SYM(Fopen, "H5Fopen"); H5SYM(Fclose, "H5Fclose"); H5SYM(Gget_num_objs, "H5Gget_num_objs");
  H5SYM(Gget_objname_by_idx, "H5Gget_objname_by_idx"); H5SYM(Gget_objtype_by_idx, "H5Gget_objtype_by_idx");
  H5SYM(Dopen2, "H5Dopen2"); H5SYM(Dclose, "H5Dclose"); H5SYM(Dget_space, "H5Dget_space"); H5SYM(Dget_type, "H5Dget_type"); H5SYM(Dread, "H5Dread");
  H5SYM(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims"); H5SYM(Sget_simple_extent_dims, "H5Sget_simple_extent_dims");
  H5SYM(Sget_simple_extent_npoints, "H5Sget_simple_extent_npoints"); H5SYM(Screate_simple, "H5Screate_simple");
  H5SYM(Sselect_hyperslab, "H5Sselect_hyperslab"); H5SYM(Sclose, "H5Sclose");
  H5SYM(Tget_class, "H5Tget_class"); H5SYM(Tget_size, "H5Tget_size"); H5SYM(Tis_variable_str, "H5Tis_variable_str"); H5SYM(Tcopy, "H5Tcopy");
  H5SYM(Tset_size, "H5Tset_size"); H5SYM(Tclose, "H5Tclose");
  H5SYM(Aexists, "H5Aexists"); H5SYM(Aopen, "H5Aopen"); H5SYM(Aclose, "H5Aclose"); H5SYM(Aget_type, "H5Aget_type"); H5SYM(Aget_space, "H5Aget_space");
  H5SYM(Aread, "H5Aread"); H5SYM(Aiterate2, "H5Aiterate2"); H5SYM(Iget_name, "H5Iget_name"); H5SYM(Oclose, "H5Oclose");
#undef H5SYM
  *(void**)&H5.Rdereference2 = dlsym(lib, "H5Rdereference2");
  *(void**)&H5.Rdereference1 = dlsym(lib, "H5Rdereference1");
  if (!H5.Rdereference2 && !H5.Rdereference1) *(void**)&H5.Rdereference1 = dlsym(lib, "H5Rdereference");      /* (1.8: the three-argument form) */
  *(void**)&H5.Dvlen_reclaim = dlsym(lib, "H5Dvlen_reclaim");
  *(void**)&H5.Eset_auto2 = dlsym(lib, "H5Eset_auto2");
  *(void**)&H5.free_memory = dlsym(lib, "H5free_memory");
  if (missing || (!H5.Rdereference2 && !H5.Rdereference1) || H5.open() < 0) { dlclose(lib); return E_NOHDF5; }
  const char* tn[] = {"H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_FLOAT_g", "H5T_NATIVE_INT_g", "H5T_NATIVE_SHORT_g", "H5T_NATIVE_SCHAR_g", "H5T_C_S1_g", "H5T_STD_REF_OBJ_g"};
  hid_t* tv[] = {&H5.t_double, &H5.t_float, &H5.t_int, &H5.t_short, &H5.t_schar, &H5.t_c_s1, &H5.t_ref_obj};
  for (int i = 0; i < 7; ++i) { const hid_t* g = (const hid_t*)dlsym(lib, tn[i]); if (!g) { dlclose(lib); return E_NOHDF5; } *tv[i] = *g; }
  if (H5.Eset_auto2) H5.Eset_auto2(0, NULL, NULL);      /* errors come back as statuses, not as text on standard error */
  H5.lib = lib;
  return 0;
}

static hid_t h5_memtype(int t) { return t == T_DOUBLE ? H5.t_double : t == T_FLOAT ? H5.t_float : t == T_INT ? H5.t_int : t == T_SHORT ? H5.t_short : H5.t_schar; }

static int h5_hidden_att(const char* name) {
  const char* hide[] = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_nc3_strict", "_NCProperties",
                        "_NCZARR_ATTR", "_Netcdf4BitGroomingOk"};
  for (size_t i = 0; i < sizeof hide / sizeof hide[0]; ++i) if (!strcmp(name, hide[i])) return 1;
  return 0;
}

/* a string attribute of `obj` (fixed or variable length) into buf; 0 if absent or not a string */
static int h5_string_att(hid_t obj, const char* name, char* buf, size_t cap) {
  buf[0] = 0;
  if (H5.Aexists(obj, name) <= 0) return 0;
  const hid_t a = H5.Aopen(obj, name, 0);
  if (a < 0) return 0;
  const hid_t t = H5.Aget_type(a);
  int ok = 0;
  if (H5.Tget_class(t) == 3) {
    if (H5.Tis_variable_str(t) > 0) {
      char* p = NULL;
      if (H5.Aread(a, t, &p) >= 0 && p) { strncpy(buf, p, cap - 1); buf[cap - 1] = 0; ok = 1; if (H5.free_memory) H5.free_memory(p); }
    } else {
      const size_t n = H5.Tget_size(t);
      char* tmp = (char*)calloc(n + 1, 1);
      if (tmp && H5.Aread(a, t, tmp) >= 0) { strncpy(buf, tmp, cap - 1); buf[cap - 1] = 0; ok = 1; }
      free(tmp);
    }
  }
  H5.Tclose(t); H5.Aclose(a);
  return ok;
}

typedef struct { Att** att; int* natt; int st; hid_t obj; } H5AttWalk;
static herr_t h5_att_cb(hid_t loc, const char* name, const void* ainfo, void* op) {
  (void)ainfo;
  H5AttWalk* w = (H5AttWalk*)op;
  if (h5_hidden_att(name) || strlen(name) >= ECNC_MAX_NAME) return 0;
  const hid_t a = H5.Aopen(loc, name, 0);
  if (a < 0) return 0;
  const hid_t t = H5.Aget_type(a), sp = H5.Aget_space(a);
  const int cls = H5.Tget_class(t);
  const size_t sz = H5.Tget_size(t);
  const long long np = H5.Sget_simple_extent_npoints(sp);
  Att x;
  memset(&x, 0, sizeof x);
  strcpy(x.name, name);
  int keep = 0;
  if (cls == 3) {      /* string: one netCDF text attribute */
    x.type = T_CHAR;
    if (H5.Tis_variable_str(t) > 0) {
      char* p = NULL;
      if (np == 1 && H5.Aread(a, t, &p) >= 0 && p) { x.n = strlen(p); x.v = malloc(x.n + 1); if (x.v) { memcpy(x.v, p, x.n + 1); keep = 1; } if (H5.free_memory) H5.free_memory(p); }
    } else if (np >= 1) {
      x.v = calloc((size_t)np * sz + 1, 1);
      if (x.v && H5.Aread(a, t, x.v) >= 0) { x.n = strnlen((const char*)x.v, (size_t)np * sz); keep = 1; }
    }
  } else if ((cls == 0 || cls == 1) && np >= 1) {
    x.type = cls == 1 ? (sz == 4 ? T_FLOAT : T_DOUBLE) : (sz == 1 ? T_BYTE : sz == 2 ? T_SHORT : sz == 4 ? T_INT : T_DOUBLE);
    x.n = (uint64_t)np;
    x.v = malloc((size_t)np * tsize(x.type));
    if (x.v && H5.Aread(a, h5_memtype(x.type), x.v) >= 0) keep = 1;
  }
  H5.Sclose(sp); H5.Tclose(t); H5.Aclose(a);
  if (!keep) { free(x.v); return 0; }
  Att* grown = (Att*)realloc(*w->att, (size_t)(*w->natt + 1) * sizeof(Att));
  if (!grown) { free(x.v); w->st = E_NOMEM; return -1; }
  *w->att = grown;
  grown[(*w->natt)++] = x;
  return 0;
}
static int h5_read_atts(hid_t obj, int* natt, Att** att) {
  H5AttWalk w = {att, natt, 0, obj};
  hsize_t idx = 0;
  *natt = 0; *att = NULL;
  /* in creation order where the file tracks it (files of libnetcdf do), otherwise in the order the attributes lie in the object header
     (H5_ITER_NATIVE over the name index): the order in which they were written, which is what nf90_inq_attname numbers */
  if (H5.Aiterate2(obj, 1 /* H5_INDEX_CRT_ORDER */, 0 /* H5_ITER_INC */, &idx, h5_att_cb, &w) < 0 && !w.st) {
    for (int i = 0; i < *natt; ++i) free((*att)[i].v);
    free(*att);
    *natt = 0; *att = NULL; idx = 0;
    H5.Aiterate2(obj, 0 /* H5_INDEX_NAME */, 2 /* H5_ITER_NATIVE */, &idx, h5_att_cb, &w);
  }
  return w.st;
}

static int ecnc_h5_open(File* f, const char* path) {
  int st = h5_load();
  if (st) return st;
  const hid_t fid = H5.Fopen(path, 0, 0);
  if (fid < 0) return E_HDFERR;
  f->h5 = fid; f->version = 6; f->recdim = -1;
  hsize_t nobj = 0;
  if (H5.Gget_num_objs(fid, &nobj) < 0) return E_HDFERR;
  f->dim = (Dim*)calloc((size_t)nobj * (ECNC_MAX_DIMS + 1) + 1, sizeof(Dim));      /* (room for anonymous dimensions of plain HDF5 files) */
  f->var = (Var*)calloc((size_t)nobj + 1, sizeof(Var));
  if (!f->dim || !f->var) return E_NOMEM;
  /* pass 1: the dimension scales, in the order of their ids where the file carries them */
  int* dimid_of = (int*)malloc(((size_t)nobj + 1) * sizeof(int));
  int* pure = (int*)calloc((size_t)nobj + 1, sizeof(int));
  char (*names)[ECNC_MAX_NAME] = (char (*)[ECNC_MAX_NAME])calloc((size_t)nobj + 1, ECNC_MAX_NAME);
  if (!dimid_of || !pure || !names) { free(dimid_of); free(pure); free(names); return E_NOMEM; }
  int nscale = 0, have_ids = 1;
  for (hsize_t i = 0; i < nobj; ++i) {
    dimid_of[i] = -1;
    if (H5.Gget_objtype_by_idx(fid, i) != 1 /* H5G_DATASET */) continue;
    if (H5.Gget_objname_by_idx(fid, i, names[i], ECNC_MAX_NAME) <= 0) { names[i][0] = 0; continue; }
    const hid_t d = H5.Dopen2(fid, names[i], 0);
    if (d < 0) { names[i][0] = 0; continue; }
    char cls[64], nm[128];
    if (h5_string_att(d, "CLASS", cls, sizeof cls) && !strcmp(cls, "DIMENSION_SCALE")) {
      dimid_of[i] = -2 - nscale++;      /* provisional: order of appearance */
      if (H5.Aexists(d, "_Netcdf4Dimid") > 0) {
        const hid_t a = H5.Aopen(d, "_Netcdf4Dimid", 0);
        int id = -1;
        if (a >= 0 && H5.Aread(a, H5.t_int, &id) >= 0 && id >= 0 && (hsize_t)id < nobj) dimid_of[i] = id; else have_ids = 0;
        if (a >= 0) H5.Aclose(a);
      } else have_ids = 0;
      if (h5_string_att(d, "NAME", nm, sizeof nm) && !strncmp(nm, "This is a netCDF dimension but not a netCDF variable", 52)) pure[i] = 1;
    }
    H5.Dclose(d);
  }
  for (hsize_t i = 0, k = 0; i < nobj; ++i) {
    if (dimid_of[i] == -1) continue;
    const int id = have_ids && dimid_of[i] >= 0 && dimid_of[i] < nscale ? dimid_of[i] : (int)k;
    if (!have_ids || dimid_of[i] < 0 || dimid_of[i] >= nscale) dimid_of[i] = id;
    ++k;
    const hid_t d = H5.Dopen2(fid, names[i], 0), sp = H5.Dget_space(d);
    hsize_t len[ECNC_MAX_DIMS + 24] = {0}, mx[ECNC_MAX_DIMS + 24] = {0};
    if (H5.Sget_simple_extent_ndims(sp) >= 1 && H5.Sget_simple_extent_ndims(sp) <= ECNC_MAX_DIMS) H5.Sget_simple_extent_dims(sp, len, mx);
    strncpy(f->dim[id].name, names[i], ECNC_MAX_NAME - 1);
    f->dim[id].len = len[0];
    if (mx[0] == (hsize_t)-1 && f->recdim < 0) f->recdim = id;      /* H5S_UNLIMITED: reported as the record dimension (its current length is its length) */
    H5.Sclose(sp); H5.Dclose(d);
  }
  f->ndim = nscale;
  /* pass 2: the variables */
  for (hsize_t i = 0; i < nobj && !st; ++i) {
    if (!names[i][0] || pure[i]) continue;
    const hid_t d = H5.Dopen2(fid, names[i], 0);
    if (d < 0) continue;
    const hid_t sp = H5.Dget_space(d), t = H5.Dget_type(d);
    const int rank = H5.Sget_simple_extent_ndims(sp), cls = H5.Tget_class(t);
    const size_t sz = H5.Tget_size(t);
    int type = 0;
    if (cls == 1) type = sz == 4 ? T_FLOAT : sz == 8 ? T_DOUBLE : 0;
    else if (cls == 0) type = sz == 1 ? T_BYTE : sz == 2 ? T_SHORT : sz == 4 ? T_INT : sz == 8 ? T_DOUBLE : 0;
    else if (cls == 3 && sz == 1 && H5.Tis_variable_str(t) <= 0) type = T_CHAR;
    if (type && rank >= 0 && rank <= ECNC_MAX_DIMS) {
      Var* v = &f->var[f->nvar];
      strncpy(v->name, names[i], ECNC_MAX_NAME - 1);
      v->type = type; v->rank = rank;
      hsize_t len[ECNC_MAX_DIMS + 1] = {0};
      if (rank) H5.Sget_simple_extent_dims(sp, len, NULL);
      int resolved = 0;
      if (dimid_of[i] >= 0 && rank == 1) { v->dimid[0] = dimid_of[i]; resolved = 1; }      /* a coordinate variable */
      else if (rank && H5.Aexists(d, "DIMENSION_LIST") > 0) {
        const hid_t a = H5.Aopen(d, "DIMENSION_LIST", 0), at = H5.Aget_type(a), as = H5.Aget_space(a);
        hvl_t refs[ECNC_MAX_DIMS];
        memset(refs, 0, sizeof refs);
        if (H5.Sget_simple_extent_npoints(as) == rank && H5.Aread(a, at, refs) >= 0) {
          resolved = 1;
          for (int k = 0; k < rank; ++k) {
            v->dimid[k] = -1;
            if (refs[k].len >= 1 && refs[k].p) {
              const hid_t o = H5.Rdereference2 ? H5.Rdereference2(fid, 0, 0 /* H5R_OBJECT */, refs[k].p) : H5.Rdereference1(fid, 0, refs[k].p);
              char on[ECNC_MAX_NAME + 2];
              if (o >= 0 && H5.Iget_name(o, on, sizeof on) > 0)
                for (int q = 0; q < f->ndim; ++q) if (!strcmp(f->dim[q].name, on[0] == '/' ? on + 1 : on)) v->dimid[k] = q;
              if (o >= 0) H5.Oclose(o);
            }
            if (v->dimid[k] < 0) resolved = 0;
          }
          if (H5.Dvlen_reclaim) H5.Dvlen_reclaim(at, as, 0, refs);
        }
        H5.Sclose(as); H5.Tclose(at); H5.Aclose(a);
      }
      if (rank && !resolved)      /* a plain HDF5 dataset: anonymous dimensions, one per distinct length (as libnetcdf's phony_dim_N) */
        for (int k = 0; k < rank; ++k) {
          int q = -1;
          for (int j = nscale; j < f->ndim; ++j) if (f->dim[j].len == len[k]) q = j;
          if (q < 0) { q = f->ndim++; snprintf(f->dim[q].name, ECNC_MAX_NAME, "phony_dim_%d", q - nscale); f->dim[q].len = len[k]; }
          v->dimid[k] = q;
        }
      v->is_rec = 0;
      st = h5_read_atts(d, &v->natt, &v->att);
      f->nvar++;
    }
    H5.Tclose(t); H5.Sclose(sp); H5.Dclose(d);
  }
  if (!st) st = h5_read_atts(fid, &f->ngatt, &f->gatt);
  free(dimid_of); free(pure); free(names);
  return st;
}

static int ecnc_h5_read(File* f, const Var* v, int memtype, void* buf, const long long* s, const long long* c) {
  const hid_t d = H5.Dopen2((hid_t)f->h5, v->name, 0);
  if (d < 0) return E_HDFERR;
  int st = 0;
  hid_t mt = h5_memtype(memtype), own = -1;
  if (memtype == T_CHAR) { own = H5.Tcopy(H5.t_c_s1); H5.Tset_size(own, 1); mt = own; }
  if (v->rank == 0) {
    if (H5.Dread(d, mt, 0, 0, 0, buf) < 0) st = E_HDFERR;
  } else {
    hsize_t start[ECNC_MAX_DIMS], count[ECNC_MAX_DIMS];
    for (int k = 0; k < v->rank; ++k) { start[k] = (hsize_t)s[k]; count[k] = (hsize_t)c[k]; }
    const hid_t fs = H5.Dget_space(d), ms = H5.Screate_simple(v->rank, count, NULL);
    if (fs < 0 || ms < 0 || H5.Sselect_hyperslab(fs, 0 /* H5S_SELECT_SET */, start, NULL, count, NULL) < 0 || H5.Dread(d, mt, ms, fs, 0, buf) < 0) st = E_HDFERR;
    if (ms >= 0) H5.Sclose(ms);
    if (fs >= 0) H5.Sclose(fs);
  }
  if (own >= 0) H5.Tclose(own);
  H5.Dclose(d);
  return st;
}

static void ecnc_h5_close(File* f) {
  if (H5.lib && f->h5 > 0) H5.Fclose((hid_t)f->h5);
  f->h5 = 0;
}

/* oracle/rt_oracle_io.c -- TEST INFRASTRUCTURE ONLY.
 * Restatement of the data formats either side of the hot path:
 *   OBJ + MTL import with Tucano's face order / material grouping, and the P3 PPM writer.
 */
#include "rt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t* v; int n, cap; int mat; } group_t;

static void group_push(group_t* g, uint32_t x) {
  if (g->n == g->cap) { g->cap = g->cap ? g->cap * 2 : 64; g->v = (uint32_t*)realloc(g->v, sizeof(uint32_t) * g->cap); }
  g->v[g->n++] = x;
}
static char* read_line(FILE* f, char** buf, size_t* cap) { /* std::getline: strips '\n' only */
  ssize_t n = getline(buf, cap, f);
  if (n < 0) return NULL;
  if (n > 0 && (*buf)[n - 1] == '\n') (*buf)[n - 1] = 0;
  return *buf;
}
static const ora_mtl DEFAULT_MTL = {{0.3f, 0.3f, 0.3f}, {0.5f, 0.5f, 0.5f}, {1.0f, 1.0f, 1.0f}, 10.0f, 0.0f, 1.0f, 0};

/* tucano/utils/mtlIO.hpp:45-125.  Tokens are split on single spaces; a trailing '\r' stays
 * part of the last token (so CRLF names still match the CRLF usemtl lines). */
static int load_mtl(const char* filename, ora_mtl** mats, char (**names)[64], int* nmats) {
  FILE* f = fopen(filename, "r");
  if (!f) return 0;
  char* line = NULL; size_t cap = 0;
  while (read_line(f, &line, &cap)) {
    if (line[0] == 0) continue;
    char* tok[8]; int nt = 0;
    char* p = line;
    while (nt < 8) { /* std::getline(ss, s, ' ') */
      tok[nt++] = p;
      char* sp = strchr(p, ' ');
      if (!sp) break;
      *sp = 0; p = sp + 1;
      if (*p == 0) break; /* a trailing delimiter yields no further token */
    }
    if (strcmp(tok[0], "#") == 0) continue;
#define TOK(i) ((i) < nt ? tok[i] : "")
    if (strcmp(tok[0], "newmtl") == 0) {
      *mats = (ora_mtl*)realloc(*mats, sizeof(ora_mtl) * (*nmats + 1));
      *names = (char (*)[64])realloc(*names, 64 * (*nmats + 1));
      (*mats)[*nmats] = DEFAULT_MTL;
      memset((*names)[*nmats], 0, 64);
      strncpy((*names)[*nmats], TOK(1), 63);
      ++*nmats;
    } else if (*nmats > 0) {
      ora_mtl* m = &(*mats)[*nmats - 1];
      if (strcmp(tok[0], "Ns") == 0) m->ns = (float)atof(TOK(1));
      else if (strcmp(tok[0], "Ka") == 0) { for (int k = 0; k < 3; ++k) m->ka[k] = (float)atof(TOK(1 + k)); }
      else if (strcmp(tok[0], "Kd") == 0) { for (int k = 0; k < 3; ++k) m->kd[k] = (float)atof(TOK(1 + k)); }
      else if (strcmp(tok[0], "Ks") == 0) { for (int k = 0; k < 3; ++k) m->ks[k] = (float)atof(TOK(1 + k)); }
      else if (strcmp(tok[0], "Ni") == 0) m->ni = (float)atof(TOK(1));
      else if (strcmp(tok[0], "d") == 0) m->d = (float)atof(TOK(1));
      else if (strcmp(tok[0], "illum") == 0) m->illum = atoi(TOK(1));
    }
#undef TOK
  }
  free(line);
  fclose(f);
  if (*nmats == 0) { /* mtlIO.hpp:114-119 */
    *mats = (ora_mtl*)realloc(*mats, sizeof(ora_mtl));
    *names = (char (*)[64])realloc(*names, 64);
    (*mats)[0] = DEFAULT_MTL; memset((*names)[0], 0, 64);
    *nmats = 1;
  }
  return 1;
}

/* tucano/utils/objimporter.hpp:83-284 + tucano/mesh.hpp:441-468 (face order) */
int ora_obj_load(const char* path, ora_mesh* out) {
  memset(out, 0, sizeof *out);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char dir[4096]; dir[0] = 0;
  const char* slash = strrchr(path, '/');
  if (slash) { size_t n = (size_t)(slash - path) + 1; if (n >= sizeof dir) n = sizeof dir - 1; memcpy(dir, path, n); dir[n] = 0; }
  float* verts = NULL; int nv = 0, vcap = 0;
  group_t* groups = (group_t*)calloc(1, sizeof(group_t)); int ng = 1;
  groups[0].mat = -1;
  ora_mtl* mats = NULL; char (*names)[64] = NULL; int nm = 0;
  int current_mat = -1;
  char* line = NULL; size_t cap = 0;
  while (read_line(f, &line, &cap)) {
    size_t len = strlen(line);
    if (len >= 6 && strncmp(line, "mtllib", 6) == 0) {
      char fn[8192];
      snprintf(fn, sizeof fn, "%s%s", dir, len > 7 ? line + 7 : "");
      char* w = fn; for (char* r = fn; *r; ++r) if (*r != '\n' && *r != '\r') *w++ = *r; *w = 0;
      load_mtl(fn, &mats, &names, &nm);
    } else if (len >= 6 && strncmp(line, "usemtl", 6) == 0) {
      if (groups[ng - 1].n != 0) {
        groups = (group_t*)realloc(groups, sizeof(group_t) * (ng + 1));
        memset(&groups[ng], 0, sizeof(group_t)); groups[ng].mat = -1; ++ng;
      }
      const char* name = len > 7 ? line + 7 : "";
      for (int i = 0; i < nm; ++i) if (strcmp(names[i], name) == 0) current_mat = i;
      groups[ng - 1].mat = current_mat;
    } else if (len >= 2 && line[0] == 'v' && line[1] == ' ') {
      if (nv == vcap) { vcap = vcap ? vcap * 2 : 256; verts = (float*)realloc(verts, sizeof(float) * 3 * vcap); }
      char* p = line + 2;
      for (int k = 0; k < 3; ++k) { char* e; float x = strtof(p, &e); if (e == p) x = 0.0f; verts[3 * nv + k] = x; p = e; }
      ++nv;
    } else if (len >= 2 && line[0] == 'f' && line[1] == ' ') {
      char* save = NULL;
      for (char* t = strtok_r(line + 2, " \t\r\n\v\f", &save); t; t = strtok_r(NULL, " \t\r\n\v\f", &save)) {
        int id = atoi(t); /* stoi of the text before the first '/' */
        group_push(&groups[ng - 1], (uint32_t)(id - 1));
      }
    }
  }
  free(line);
  fclose(f);
  int ntris = 0;
  for (int g = 0; g < ng; ++g) ntris += groups[g].n / 3;
  out->nverts = nv; out->verts = verts ? verts : (float*)malloc(4);
  out->ntris = ntris;
  out->idx = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (ntris > 0 ? ntris : 1));
  out->tri_mat = (int32_t*)malloc(sizeof(int32_t) * (ntris > 0 ? ntris : 1));
  int k = 0;
  for (int g = 0; g < ng; ++g) {
    for (int i = 0; i + 2 < groups[g].n; i += 3) {
      out->idx[3 * k] = groups[g].v[i]; out->idx[3 * k + 1] = groups[g].v[i + 1]; out->idx[3 * k + 2] = groups[g].v[i + 2];
      out->tri_mat[k] = groups[g].mat;
      ++k;
    }
    free(groups[g].v);
  }
  free(groups);
  out->nmats = nm; out->mats = mats; out->mat_names = names;
  return 0;
}
void ora_mesh_free(ora_mesh* m) {
  free(m->verts); free(m->idx); free(m->tri_mat); free(m->mats); free(m->mat_names);
  memset(m, 0, sizeof *m);
}

/* tucano/utils/ppmIO.hpp:130-151.  The reference stores pixel (x,y) at data[x][y] and the
 * writer emits data[i][j] at row j, column i, so for the square images it supports the file
 * is the ordinary row-major image. */
int ora_ppm_write(const char* path, const float* rgb, int w, int h, int reference_indexing) {
  (void)reference_indexing;
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "P3\n%d %d\n255\n", w, h);
  for (int j = 0; j < h; ++j) {
    for (int i = 0; i < w; ++i) {
      const float* c = rgb + 3 * ((size_t)j * w + i);
      int r = (int)(255 * c[0]), g = (int)(255 * c[1]), b = (int)(255 * c[2]);
      fprintf(f, "%d %d %d ", r < 255 ? r : 255, g < 255 ? g : 255, b < 255 ? b : 255);
    }
    fprintf(f, "\n");
  }
  fclose(f);
  return 0;
}

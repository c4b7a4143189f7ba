/*
 * llvmpipe_ref.c -- TEST INFRASTRUCTURE (oracle side). Not part of the product path.
 *
 * Runs the reference's own compute shader, read at run time from its mounted
 * location (/root/reference/src/shaders/svotrace.comp -- never copied into this
 * repository), on the CPU through Mesa llvmpipe.  There is no X server, EGL or
 * OSMesa in the build container, so the GL context is obtained by talking to
 * swrast_dri.so through the raw DRI "swrast loader" interface.
 *
 * It mirrors what the reference host does around the dispatch:
 *   Main.java:59-127   textures rgba8 (image unit 0) + r32f (unit 1), SSBO on binding 7
 *   Main.java:267-285  uniforms 8 (camPos), 1..4 (l1,l2,r1,r2), 5 frameNumber,
 *                      6 renderMode, 9 bufferEnd, 11 useBeamOptimization
 *   Renderer.java:118-121  glDispatchCompute + glMemoryBarrier
 *
 * Usage:  llvmpipe_ref <shader.comp> [raw]  < jobfile      (raw: a probe shader with the same bindings, no patches)
 * Job file (one command per line):
 *   pool <file>                 raw SVO byte pool (T1 layout)
 *   pad <bytes>                 zero bytes behind the pools that follow (the reference's buffer is far larger
 *                               than the bytes in use); default 64
 *   size <W> <H>
 *   cam <15 hex u32>            bit patterns of pos,l1,l2,r1,r2 floats
 *   frame <n>   mode <m>
 *   ptrpatch <0|1>              1: apply an IN-MEMORY patch to the source text so
 *                               the first cast's hit pointer is exported through an
 *                               extra r32ui image (the live shader has that store
 *                               commented out, svotrace.comp:728)
 *   accum <0|1>                 1: use the program whose dormant cross-frame accumulation
 *                               (svotrace.comp:712-719, commented out) is switched on in memory
 *   keep <0|1>                  1: the images persist from render to render, as in Main.java
 *   fresh                       drop the persistent images (the next render starts from cleared ones)
 *   bounces <n>                 path segments of renderMode 0: the literal loop bound 2 of svotrace.comp:444 becomes n
 *   mirror <0|1>                1: the commented-out material test of svotrace.comp:500-504 is switched on in memory
 *                               (value 1 scatters, every other value reflects) and the unconditional scatter of :506 off
 *   render <prefix>             writes <prefix>.rgba  <prefix>.depth  [<prefix>.ptr]
 *
 *
 * Second mode -- the world builder's first stage:  llvmpipe_ref <chunkgen-heightmap.comp> chunkgen  < jobfile
 * runs the reference's voxelisation shader (height map + material map -> a dense chunk of voxels) the way
 * src/tests/WorldGenerator.java:24-37 and Octree.constructCompleteOctree (Octree.java:216, 229, 274-287) drive it:
 * r8i 3-D image on unit 3, r16ui height image on unit 4 (glTexSubImage2D GL_RED_INTEGER / GL_UNSIGNED_SHORT), r8i
 * material image on unit 5 (GL_RED_INTEGER / GL_BYTE), chunk origin in uniforms 1-3, chunk / 8 work groups per axis,
 * read back with glGetTexImage(GL_TEXTURE_3D, GL_RED_INTEGER, GL_BYTE).
 *   maps <n> <height file: n*n u16, [z][x]> <material file: n*n u8>
 *   chunk <size> <ox> <oy> <oz> <out file>        size^3 bytes, [z][y][x]
 *
 * Build: see oracle/Makefile (output goes to oracle/_ref/, git-ignored).
 */
#define _GNU_SOURCE
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
  (void)d; (void)p; *x = *y = 0; *w = *h = 16;
}
static void putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
  (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static const __DRIswrastLoaderExtension swl = {
    .base = {__DRI_SWRAST_LOADER, 1},
    .getDrawableInfo = getDrawableInfo,
    .putImage = putImage,
    .getImage = getImage};
static const __DRIextension *loader_exts[] = {&swl.base, NULL};

static void *(*gpa)(const char *);
#define GLF(type, name) static type name;
#define GLLOAD(type, name)                                   \
  do {                                                       \
    name = (type)gpa(#name);                                 \
    if (!name) { fprintf(stderr, "missing %s\n", #name); exit(2); } \
  } while (0)

GLF(PFNGLGETSTRINGPROC, glGetString)
GLF(PFNGLCREATESHADERPROC, glCreateShader)
GLF(PFNGLSHADERSOURCEPROC, glShaderSource)
GLF(PFNGLCOMPILESHADERPROC, glCompileShader)
GLF(PFNGLGETSHADERIVPROC, glGetShaderiv)
GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog)
GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram)
GLF(PFNGLATTACHSHADERPROC, glAttachShader)
GLF(PFNGLLINKPROGRAMPROC, glLinkProgram)
GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv)
GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog)
GLF(PFNGLUSEPROGRAMPROC, glUseProgram)
GLF(PFNGLGENTEXTURESPROC, glGenTextures)
GLF(PFNGLDELETETEXTURESPROC, glDeleteTextures)
GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture)
GLF(PFNGLBINDTEXTUREPROC, glBindTexture)
GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri)
GLF(PFNGLTEXSTORAGE2DPROC, glTexStorage2D)
GLF(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture)
GLF(PFNGLGENBUFFERSPROC, glGenBuffers)
GLF(PFNGLDELETEBUFFERSPROC, glDeleteBuffers)
GLF(PFNGLBINDBUFFERPROC, glBindBuffer)
GLF(PFNGLBUFFERDATAPROC, glBufferData)
GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase)
GLF(PFNGLUNIFORM3FVPROC, glUniform3fv)
GLF(PFNGLUNIFORM1IPROC, glUniform1i)
GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute)
GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier)
GLF(PFNGLFINISHPROC, glFinish)
GLF(PFNGLGETTEXIMAGEPROC, glGetTexImage)
GLF(PFNGLGETERRORPROC, glGetError)
GLF(PFNGLPIXELSTOREIPROC, glPixelStorei)
GLF(PFNGLTEXSTORAGE3DPROC, glTexStorage3D)
GLF(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D)

static void load_gl(void) {
  GLLOAD(PFNGLGETSTRINGPROC, glGetString);
  GLLOAD(PFNGLCREATESHADERPROC, glCreateShader);
  GLLOAD(PFNGLSHADERSOURCEPROC, glShaderSource);
  GLLOAD(PFNGLCOMPILESHADERPROC, glCompileShader);
  GLLOAD(PFNGLGETSHADERIVPROC, glGetShaderiv);
  GLLOAD(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
  GLLOAD(PFNGLCREATEPROGRAMPROC, glCreateProgram);
  GLLOAD(PFNGLATTACHSHADERPROC, glAttachShader);
  GLLOAD(PFNGLLINKPROGRAMPROC, glLinkProgram);
  GLLOAD(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
  GLLOAD(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
  GLLOAD(PFNGLUSEPROGRAMPROC, glUseProgram);
  GLLOAD(PFNGLGENTEXTURESPROC, glGenTextures);
  GLLOAD(PFNGLDELETETEXTURESPROC, glDeleteTextures);
  GLLOAD(PFNGLACTIVETEXTUREPROC, glActiveTexture);
  GLLOAD(PFNGLBINDTEXTUREPROC, glBindTexture);
  GLLOAD(PFNGLTEXPARAMETERIPROC, glTexParameteri);
  GLLOAD(PFNGLTEXSTORAGE2DPROC, glTexStorage2D);
  GLLOAD(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
  GLLOAD(PFNGLGENBUFFERSPROC, glGenBuffers);
  GLLOAD(PFNGLDELETEBUFFERSPROC, glDeleteBuffers);
  GLLOAD(PFNGLBINDBUFFERPROC, glBindBuffer);
  GLLOAD(PFNGLBUFFERDATAPROC, glBufferData);
  GLLOAD(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
  GLLOAD(PFNGLUNIFORM3FVPROC, glUniform3fv);
  GLLOAD(PFNGLUNIFORM1IPROC, glUniform1i);
  GLLOAD(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
  GLLOAD(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
  GLLOAD(PFNGLFINISHPROC, glFinish);
  GLLOAD(PFNGLGETTEXIMAGEPROC, glGetTexImage);
  GLLOAD(PFNGLGETERRORPROC, glGetError);
  GLLOAD(PFNGLPIXELSTOREIPROC, glPixelStorei);
  GLLOAD(PFNGLTEXSTORAGE3DPROC, glTexStorage3D);
  GLLOAD(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D);
}

static char *read_file(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *b = malloc((size_t)n + 1);
  if (fread(b, 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(2); }
  fclose(f);
  b[n] = 0;
  if (len) *len = (size_t)n;
  return b;
}

/* replace every occurrence of `a` by `b` inside [from, to) markers; returns new string */
static char *replace_all(const char *src, const char *a, const char *b, int *count) {
  size_t la = strlen(a), lb = strlen(b), n = 0;
  for (const char *p = src; (p = strstr(p, a)); p += la) n++;
  char *out = malloc(strlen(src) + n * (lb > la ? lb - la : 0) + 1), *o = out;
  const char *p = src, *q;
  while ((q = strstr(p, a))) {
    memcpy(o, p, (size_t)(q - p)); o += q - p;
    memcpy(o, b, lb); o += lb;
    p = q + la;
  }
  strcpy(o, p);
  if (count) *count = (int)n;
  return out;
}

/*
 * In-memory instrumentation of the source text (the file on disk is untouched):
 * export, through a new r32ui image on unit 3, the byte offset of the node the FIRST
 * intersectOctree cast of the pixel hit (0 on a miss -- same convention as
 * svobeam.comp:538,553), and the value / leafMask field of that node.
 */
static char *apply_ptr_patch(const char *src) {
  int n;
  char *s1 = replace_all(src,
      "layout(binding = 2, r32f) uniform image2D beambufferImage;",
      "layout(binding = 2, r32f) uniform image2D beambufferImage;\n"
      "layout(binding = 3, rgba32ui) uniform uimage2D pointerbufferImage;\n"
      "uint g_casts = 0u; uvec4 g_first = uvec4(0u);", &n);
  if (n != 1) { fprintf(stderr, "patch anchor 1 not found\n"); exit(2); }
  /* all early `return false;` in the file live inside intersectOctree */
  char *s2 = replace_all(s1, "return false;", "{ g_casts++; return false; }", &n);
  if (n != 2) { fprintf(stderr, "patch anchor 2: %d\n", n); exit(2); }
  char *s3 = replace_all(s2, "return scale < MAX_SCALE && t_min <= t_max;",
      "{ bool ok_ = scale < MAX_SCALE && t_min <= t_max;\n"
      "  if(g_casts == 0u) g_first = uvec4(ok_ ? res.pointer : 0u, uint(targetNode.value), uint(targetNode.leafMask), (uint(MAX_SCALE - scale) << 16) | iter);\n"
      "  g_casts++; return ok_; }", &n);
  if (n != 1) { fprintf(stderr, "patch anchor 3: %d\n", n); exit(2); }
  char *s4 = replace_all(s3, "// imageStore(pointerbufferImage, px, uvec4(iter, 0, 0, 0));",
      "imageStore(pointerbufferImage, px, g_first);", &n);
  if (n != 1) { fprintf(stderr, "patch anchor 4: %d\n", n); exit(2); }
  free(s1); free(s2); free(s3);
  return s4;
}

/*
 * In-memory activation of the reference's dormant cross-frame accumulation: the block at svotrace.comp:712-719
 * (`if(frameNumber > 1){ lastcolor = imageLoad(...); finalcolor = (frameNumber * lastcolor + finalcolor) /
 * (frameNumber + 1) ... }`) is commented out line by line; this strips the leading "// " of exactly those lines.
 * Addressed by line number (no shader text lives here); the first line is checked for the frameNumber test.
 */
static char *apply_accum_patch(const char *src, int first_line, int last_line) {
  char *out = malloc(strlen(src) + 1), *o = out;
  int line = 1, checked = 0;
  const char *p = src;
  while (*p) {
    const char *e = strchr(p, '\n');
    size_t n = e ? (size_t)(e - p) + 1 : strlen(p);
    if (line >= first_line && line <= last_line) {
      const char *c = p;
      while (c < p + n && (*c == ' ' || *c == '\t')) c++;
      if (c + 2 <= p + n && c[0] == '/' && c[1] == '/') {
        if (line == first_line) {
          char tmp[256];
          size_t m = n < 255 ? n : 255;
          memcpy(tmp, p, m); tmp[m] = 0;
          checked = strstr(tmp, "frameNumber > 1") != NULL;
        }
        memcpy(o, p, (size_t)(c - p)); o += c - p;       /* indentation */
        c += 2;
        memcpy(o, c, (size_t)(p + n - c)); o += p + n - c;
        p += n; line++;
        continue;
      }
      fprintf(stderr, "accum patch: line %d is not a comment\n", line);
      exit(2);
    }
    memcpy(o, p, n); o += n;
    p += n; line++;
  }
  *o = 0;
  if (!checked) { fprintf(stderr, "accum patch: line %d is not the frameNumber test\n", first_line); exit(2); }
  return out;
}

/*
 * Two more in-memory switches for code the reference carries but does not run (no shader text lives here either):
 *   bounces N : the literal bound of the path loop of renderMode 0 (`i < 2`, svotrace.comp:444) becomes N;
 *   mirror    : the commented-out material test of svotrace.comp:500-504 (value 1 scatters, every other value reflects:
 *               dir - 2 dot(dir, normal) normal) is uncommented and the unconditional scatter of :506 commented out.
 * Line counts do not change, so the line-addressed patches compose.
 */
static char *toggle_lines(const char *src, int first_line, int last_line, int uncomment, const char *must_contain) {
  char *out = malloc(strlen(src) + 4 * (size_t)(last_line - first_line + 2) + 1), *o = out;
  int line = 1, checked = 0;
  const char *p = src;
  while (*p) {
    const char *e = strchr(p, '\n');
    size_t n = e ? (size_t)(e - p) + 1 : strlen(p);
    if (line >= first_line && line <= last_line) {
      if (line == first_line) {
        char tmp[512];
        size_t m = n < 511 ? n : 511;
        memcpy(tmp, p, m); tmp[m] = 0;
        checked = strstr(tmp, must_contain) != NULL;
      }
      const char *c = p;
      while (c < p + n && (*c == ' ' || *c == '\t')) c++;
      if (uncomment) {
        if (!(c + 2 <= p + n && c[0] == '/' && c[1] == '/')) { fprintf(stderr, "line %d is not a comment\n", line); exit(2); }
        memcpy(o, p, (size_t)(c - p)); o += c - p;
        c += 2;
        memcpy(o, c, (size_t)(p + n - c)); o += p + n - c;
      } else {
        memcpy(o, "//", 2); o += 2;
        memcpy(o, p, n); o += n;
      }
      p += n; line++;
      continue;
    }
    memcpy(o, p, n); o += n;
    p += n; line++;
  }
  *o = 0;
  if (!checked) { fprintf(stderr, "line %d does not contain \"%s\"\n", first_line, must_contain); exit(2); }
  return out;
}

static char *apply_variant(const char *src, int bounces, int mirror) {
  char *s = strdup(src);
  if (bounces != 2) {
    char with[64];
    int n;
    snprintf(with, sizeof with, "for(int i=0; i < %d; i++){", bounces);
    char *t = replace_all(s, "for(int i=0; i < 2; i++){", with, &n);
    if (n != 1) { fprintf(stderr, "bounce loop anchor: %d\n", n); exit(2); }
    free(s); s = t;
  }
  if (mirror) {
    char *t = toggle_lines(s, 500, 504, 1, "res.value == 1");
    free(s);
    s = toggle_lines(t, 506, 506, 0, "newdir = normalize");
    free(t);
  }
  return s;
}

static GLuint build_program(const char *src) {
  GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
  glShaderSource(sh, 1, &src, NULL);
  glCompileShader(sh);
  GLint ok = 0;
  glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[8192];
    glGetShaderInfoLog(sh, sizeof log, NULL, log);
    fprintf(stderr, "compile failed:\n%s\n", log);
    exit(3);
  }
  GLuint prog = glCreateProgram();
  glAttachShader(prog, sh);
  glLinkProgram(prog);
  glGetProgramiv(prog, GL_LINK_STATUS, &ok);
  if (!ok) {
    char log[8192];
    glGetProgramInfoLog(prog, sizeof log, NULL, log);
    fprintf(stderr, "link failed:\n%s\n", log);
    exit(3);
  }
  return prog;
}

static void write_file(const char *prefix, const char *ext, const void *data, size_t n) {
  char path[4096];
  snprintf(path, sizeof path, "%s%s", prefix, ext);
  FILE *f = fopen(path, "wb");
  if (!f) { perror(path); exit(2); }
  fwrite(data, 1, n, f);
  fclose(f);
}

/* chunkgen-heightmap.comp, driven as WorldGenerator.java / Octree.constructCompleteOctree drive it (see the header) */
static int chunkgen_jobs(GLuint prog) {
  GLuint maps[2] = {0, 0};
  int have_maps = 0;
  char line[8192];
  glPixelStorei(GL_PACK_ALIGNMENT, 1);
  glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
  while (fgets(line, sizeof line, stdin)) {
    char cmd[64], a[4096], b[4096];
    if (sscanf(line, "%63s", cmd) != 1) continue;
    if (!strcmp(cmd, "maps")) {
      int n = 0;
      if (sscanf(line, "%*s %d %4095s %4095s", &n, a, b) != 3 || n <= 0) { fprintf(stderr, "maps <n> <height> <material>\n"); return 2; }
      size_t lh = 0, lm = 0;
      char *hm = read_file(a, &lh), *mm = read_file(b, &lm);
      if (lh != (size_t)n * n * 2 || lm != (size_t)n * n) { fprintf(stderr, "map sizes\n"); return 2; }
      if (have_maps) glDeleteTextures(2, maps);
      glGenTextures(2, maps);
      glActiveTexture(GL_TEXTURE4);                       /* Renderer.add2DTexture(4, GL_R16UI, ...) + buffer2DTexture */
      glBindTexture(GL_TEXTURE_2D, maps[0]);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage2D(GL_TEXTURE_2D, 1, GL_R16UI, n, n);
      glBindImageTexture(4, maps[0], 0, GL_TRUE, 0, GL_READ_WRITE, GL_R16UI);
      glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, n, n, GL_RED_INTEGER, GL_UNSIGNED_SHORT, hm);
      glActiveTexture(GL_TEXTURE5);                       /* Renderer.add2DTexture(5, GL_R8I, ...) + buffer2DTexture */
      glBindTexture(GL_TEXTURE_2D, maps[1]);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage2D(GL_TEXTURE_2D, 1, GL_R8I, n, n);
      glBindImageTexture(5, maps[1], 0, GL_TRUE, 0, GL_READ_WRITE, GL_R8I);
      glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, n, n, GL_RED_INTEGER, GL_BYTE, mm);
      free(hm); free(mm);
      have_maps = 1;
    } else if (!strcmp(cmd, "chunk")) {
      int c = 0, ox = 0, oy = 0, oz = 0;
      if (sscanf(line, "%*s %d %d %d %d %4095s", &c, &ox, &oy, &oz, a) != 5 || c < 8 || (c & 7) || !have_maps) {
        fprintf(stderr, "chunk <size %% 8 == 0> <ox> <oy> <oz> <out> (after maps)\n");
        return 2;
      }
      GLuint vox = 0;
      glGenTextures(1, &vox);
      glActiveTexture(GL_TEXTURE3);                       /* Renderer.add3DTexture(3, GL_R8I, c, c, c) */
      glBindTexture(GL_TEXTURE_3D, vox);
      glTexParameteri(GL_TEXTURE_3D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage3D(GL_TEXTURE_3D, 1, GL_R8I, c, c, c);
      glBindImageTexture(3, vox, 0, GL_TRUE, 0, GL_WRITE_ONLY, GL_R8I);
      glUseProgram(prog);
      glUniform1i(1, ox);
      glUniform1i(2, oy);
      glUniform1i(3, oz);
      glDispatchCompute((GLuint)(c / 8), (GLuint)(c / 8), (GLuint)(c / 8));
      glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
      glFinish();
      size_t nb = (size_t)c * c * c;
      void *out = malloc(nb);
      glGetTexImage(GL_TEXTURE_3D, 0, GL_RED_INTEGER, GL_BYTE, out);   /* Renderer.get3DTextureData */
      write_file(a, "", out, nb);
      free(out);
      GLenum e = glGetError();
      if (e) fprintf(stderr, "GL error 0x%x after chunk %s\n", e, a);
      glDeleteTextures(1, &vox);
      fprintf(stderr, "chunk %d^3 at (%d, %d, %d) -> %s\n", c, ox, oy, oz, a);
    }
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s shader.comp < jobs\n", argv[0]); return 2; }
  void *h = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen swrast: %s\n", dlerror()); return 2; }
  const __DRIextension **(*getexts)(void) =
      (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
  if (!getexts) { fprintf(stderr, "no __driDriverGetExtensions_swrast\n"); return 2; }
  const __DRIextension **exts = getexts();
  const __DRIcoreExtension *core = NULL;
  const __DRIswrastExtension *sw = NULL;
  for (int i = 0; exts[i]; i++) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *)exts[i];
  }
  if (!core || !sw) { fprintf(stderr, "DRI core/swrast ext missing\n"); return 2; }
  const __DRIconfig **cfg;
  __DRIscreen *scr = sw->createNewScreen2(0, loader_exts, exts, &cfg, NULL);
  if (!scr) { fprintf(stderr, "createNewScreen2 failed\n"); return 2; }
  uint32_t at[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
  unsigned err = 0;
  __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, cfg[0], NULL, 2, at, &err, NULL);
  if (!ctx) { fprintf(stderr, "createContextAttribs failed %u\n", err); return 2; }
  __DRIdrawable *dr = sw->createNewDrawable(scr, cfg[0], NULL);
  if (!core->bindContext(ctx, dr, dr)) { fprintf(stderr, "bindContext failed\n"); return 2; }
  void *ga = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!ga) { fprintf(stderr, "dlopen glapi: %s\n", dlerror()); return 2; }
  gpa = (void *(*)(const char *))dlsym(ga, "_glapi_get_proc_address");
  if (!gpa) { fprintf(stderr, "no _glapi_get_proc_address\n"); return 2; }
  load_gl();
  fprintf(stderr, "GL_VERSION %s | %s\n", glGetString(GL_VERSION), glGetString(GL_RENDERER));

  char *src = read_file(argv[1], NULL);
  if (argc > 2 && !strcmp(argv[2], "chunkgen")) return chunkgen_jobs(build_program(src));
  /* "raw": a probe shader of our own with the same bindings (tools/probes/), run as it is -- no in-memory patches */
  const int raw = argc > 2 && !strcmp(argv[2], "raw");
  GLuint prog_plain = build_program(src);
  GLuint prog_patched = raw ? prog_plain : build_program(apply_ptr_patch(src));
  GLuint prog_accum = raw ? prog_plain : build_program(apply_accum_patch(src, 712, 719));

  int W = 256, H = 256, frame = 2, mode = 2, ptrpatch = 0, accum = 0, keep = 0, have_tex = 0, texW = 0, texH = 0;
  GLuint tex[3] = {0, 0, 0};
  float cam[15] = {1.5f, 1.5f, 2.0f, -1.6f, -0.9f, -1, -1.6f, 0.9f, -1, 1.6f, -0.9f, -1, 1.6f, 0.9f, -1};
  GLuint ssbo = 0;
  size_t pool_len = 0, pad_bytes = 0;
  int var_bounces = 2, var_mirror = 0;
  char line[8192];
  glPixelStorei(GL_PACK_ALIGNMENT, 1);
  while (fgets(line, sizeof line, stdin)) {
    char cmd[64], arg[4096];
    if (sscanf(line, "%63s", cmd) != 1) continue;
    if (!strcmp(cmd, "pool")) {
      sscanf(line, "%*s %4095s", arg);
      char *pool = read_file(arg, &pool_len);
      /* the reference over-allocates its buffer (Octree.java:63-67); pad with zeros so
         the shader's dword reads past the last record stay inside the SSBO */
      size_t padded = ((pool_len + 3) & ~(size_t)3) + 64 + pad_bytes;
      char *buf = calloc(1, padded);
      memcpy(buf, pool, pool_len);
      if (ssbo) glDeleteBuffers(1, &ssbo);
      glGenBuffers(1, &ssbo);
      glBindBuffer(GL_SHADER_STORAGE_BUFFER, ssbo);
      glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)padded, buf, GL_DYNAMIC_DRAW);
      glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 7, ssbo);
      free(buf); free(pool);
    } else if (!strcmp(cmd, "pad")) {          /* zero bytes behind the pools that follow: the reference's buffer is far
                                                  larger than the bytes in use (Octree.java:63-67, Renderer.java:101-113) */
      unsigned long v = 0;
      sscanf(line, "%*s %lu", &v);
      pad_bytes = (size_t)v;
    } else if (!strcmp(cmd, "bounces") || !strcmp(cmd, "mirror")) {   /* rebuild the programs from the switched source */
      int v = 0;
      sscanf(line, "%*s %d", &v);
      if (!strcmp(cmd, "bounces")) var_bounces = v; else var_mirror = v;
      if (raw) { fprintf(stderr, "variants need the reference shader\n"); return 2; }
      char *vs = apply_variant(src, var_bounces, var_mirror);
      prog_plain = build_program(vs);
      prog_patched = build_program(apply_ptr_patch(vs));
      prog_accum = build_program(apply_accum_patch(vs, 712, 719));
      free(vs);
    } else if (!strcmp(cmd, "size")) {
      sscanf(line, "%*s %d %d", &W, &H);
    } else if (!strcmp(cmd, "cam")) {
      char *p = line + 3;
      for (int i = 0; i < 15; i++) {
        uint32_t u = (uint32_t)strtoul(p, &p, 16);
        memcpy(&cam[i], &u, 4);
      }
    } else if (!strcmp(cmd, "frame")) {
      sscanf(line, "%*s %d", &frame);
    } else if (!strcmp(cmd, "mode")) {
      sscanf(line, "%*s %d", &mode);
    } else if (!strcmp(cmd, "ptrpatch")) {
      sscanf(line, "%*s %d", &ptrpatch);
    } else if (!strcmp(cmd, "accum")) {        /* 1: the program with svotrace.comp:712-719 active */
      sscanf(line, "%*s %d", &accum);
    } else if (!strcmp(cmd, "keep")) {         /* 1: the images persist from render to render (Main.java allocates them once) */
      sscanf(line, "%*s %d", &keep);
    } else if (!strcmp(cmd, "fresh")) {        /* drop the persistent images: the next render starts from cleared ones */
      if (have_tex) { glDeleteTextures(3, tex); have_tex = 0; }
    } else if (!strcmp(cmd, "render")) {
      sscanf(line, "%*s %4095s", arg);
      const int reuse = keep && have_tex && texW == W && texH == H;
      if (have_tex && !reuse) { glDeleteTextures(3, tex); have_tex = 0; }
      if (!reuse) {
      glGenTextures(3, tex);
      glActiveTexture(GL_TEXTURE0);
      glBindTexture(GL_TEXTURE_2D, tex[0]);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA8, W, H);
      glBindImageTexture(0, tex[0], 0, GL_TRUE, 0, GL_READ_WRITE, GL_RGBA8);
      glActiveTexture(GL_TEXTURE1);
      glBindTexture(GL_TEXTURE_2D, tex[1]);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage2D(GL_TEXTURE_2D, 1, GL_R32F, W, H);
      glBindImageTexture(1, tex[1], 0, GL_TRUE, 0, GL_READ_WRITE, GL_R32F);
      glActiveTexture(GL_TEXTURE3);
      glBindTexture(GL_TEXTURE_2D, tex[2]);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
      glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
      glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32UI, W, H);
      glBindImageTexture(3, tex[2], 0, GL_TRUE, 0, GL_READ_WRITE, GL_RGBA32UI);
      have_tex = 1; texW = W; texH = H;
      }

      glUseProgram(accum ? prog_accum : (ptrpatch ? prog_patched : prog_plain));
      glUniform3fv(8, 1, cam + 0);
      glUniform3fv(1, 1, cam + 3);
      glUniform3fv(2, 1, cam + 6);
      glUniform3fv(3, 1, cam + 9);
      glUniform3fv(4, 1, cam + 12);
      glUniform1i(5, frame);
      glUniform1i(6, mode);
      glUniform1i(9, (GLint)pool_len);
      glUniform1i(11, 0);
      glDispatchCompute((GLuint)((W + 7) / 8), (GLuint)((H + 7) / 8), 1);
      glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
      glFinish();

      size_t npx = (size_t)W * (size_t)H;
      void *rgba = malloc(npx * 4), *depth = malloc(npx * 4), *ptr = malloc(npx * 16);
      glActiveTexture(GL_TEXTURE0);
      glBindTexture(GL_TEXTURE_2D, tex[0]);
      glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, rgba);
      glActiveTexture(GL_TEXTURE1);
      glBindTexture(GL_TEXTURE_2D, tex[1]);
      glGetTexImage(GL_TEXTURE_2D, 0, GL_RED, GL_FLOAT, depth);
      write_file(arg, ".rgba", rgba, npx * 4);
      write_file(arg, ".depth", depth, npx * 4);
      if (ptrpatch && !accum) {
        glActiveTexture(GL_TEXTURE3);
        glBindTexture(GL_TEXTURE_2D, tex[2]);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA_INTEGER, GL_UNSIGNED_INT, ptr);
        write_file(arg, ".ptr", ptr, npx * 16);
      }
      GLenum e = glGetError();
      if (e) fprintf(stderr, "GL error 0x%x after render %s\n", e, arg);
      free(rgba); free(depth); free(ptr);
      if (!keep) { glDeleteTextures(3, tex); have_tex = 0; }
      fprintf(stderr, "rendered %s %dx%d mode %d frame %d patch %d accum %d\n", arg, W, H, mode, frame, ptrpatch, accum);
    }
  }
  return 0;
}

/* TEST INFRASTRUCTURE, build container only (never loaded by the product, never needed on the GPU box).
 *
 * A windowless OpenGL context on Mesa's software rasteriser (llvmpipe), made by driving the image's own
 * /usr/lib/x86_64-linux-gnu/dri/swrast_dri.so through the DRI software-rasteriser loader interface declared in the
 * image's <GL/internal/dri_interface.h> -- the interface libGLX_mesa / OSMesa use internally; there is no X server,
 * EGL or OSMesa in the image, but this needs none of them.  oracle/gl_ref.py uses the context to RUN the reference's GLSL
 * files where they lie under /root/reference/glsl (compiled by Mesa's GLSL compiler, sampled by its texture units) and
 * freezes the results as tests/golden/gl_*.npz.
 *
 *   glctx_create(compat, major, minor) -> 0 on success; the context stays current on the calling thread
 *   glctx_proc(name)                   -> GL entry point (libglapi's dispatch stubs)
 *   glctx_destroy()
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <GL/gl.h>
#include <GL/internal/dri_interface.h>

static void drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv) {
    (void)d; (void)priv;
    *x = *y = 0;
    *w = *h = 16;
}
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv;
}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv) {
    (void)d; (void)x; (void)y; (void)priv;
    memset(data, 0, (size_t)w * (size_t)h * 4);
}
static void put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *priv) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)priv;
}
static void get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *priv) {
    (void)d; (void)x; (void)y; (void)w; (void)priv;
    memset(data, 0, (size_t)stride * (size_t)h);
}

static const __DRIswrastLoaderExtension loader_ext = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = drawable_info,
    .putImage = put_image,
    .getImage = get_image,
    .putImage2 = put_image2,
    .getImage2 = get_image2,
};
static const __DRIextension *loader_exts[] = {&loader_ext.base, NULL};

static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_get_proc)(const char *);

int glctx_create(int compat, int major, int minor) {
    if (g_ctx) return 0;
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "glctx: %s\n", dlerror()); return 1; }
    const __DRIextension **(*get_exts)(void) = (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) return 2;
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) return 3;
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) return 4;
    /* an RGBA8 + depth 24 configuration if there is one (only FBOs are rendered to; the drawable is a dummy) */
    const __DRIconfig *cfg = configs[0];
    for (int i = 0; configs[i]; i++) {
        unsigned r = 0, a = 0, z = 0, db = 0;
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &z);
        g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && a == 8 && z == 24 && !db) { cfg = configs[i]; break; }
    }
    unsigned err = 0;
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)major, __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t)minor};
    g_ctx = g_swrast->createContextAttribs(g_screen, compat ? __DRI_API_OPENGL : __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { fprintf(stderr, "glctx: createContextAttribs error %u\n", err); return 5; }
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) return 6;
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return 7;
    void *api = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!api) return 8;
    g_get_proc = (void *(*)(const char *))dlsym(api, "_glapi_get_proc_address");
    return g_get_proc ? 0 : 9;
}

void *glctx_proc(const char *name) { return g_get_proc ? g_get_proc(name) : NULL; }

void glctx_destroy(void) {
    if (!g_ctx) return;
    g_core->unbindContext(g_ctx);
    g_core->destroyDrawable(g_draw);
    g_core->destroyContext(g_ctx);
    g_core->destroyScreen(g_screen);
    g_ctx = NULL;
}

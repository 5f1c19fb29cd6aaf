/* minifb_stub.c -- see MiniFB.h in this directory. */
#include "MiniFB.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

int clw_host_write_png(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height);

struct mfb_window {
    unsigned width, height;
    mfb_keyboard_func on_key;
    long frames_left, frames_done;
    const char* keys;
    size_t key_pos;
    double t_first, t_last;
    uint32_t* last;
};
struct mfb_timer { double t0; };

static double now(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

struct mfb_window* mfb_open_ex(const char* title, unsigned width, unsigned height, unsigned flags) {
    (void)title; (void)flags;
    struct mfb_window* w = (struct mfb_window*)calloc(1, sizeof *w);
    const char* f = getenv("MFB_STUB_FRAMES");
    w->width = width; w->height = height;
    w->frames_left = f ? atol(f) : 100;
    w->keys = getenv("MFB_STUB_KEYS");
    w->last = (uint32_t*)malloc((size_t)width * height * 4);
    return w;
}

void mfb_set_keyboard_callback(struct mfb_window* window, mfb_keyboard_func callback) { window->on_key = callback; }

static mfb_key decode(char c) {
    switch (c) {
        case 'u': return KB_KEY_UP;   case 'd': return KB_KEY_DOWN;
        case 'l': return KB_KEY_LEFT; case 'r': return KB_KEY_RIGHT;
        case 'W': return KB_KEY_W;    case 'S': return KB_KEY_S;
        case 'A': return KB_KEY_A;    case 'D': return KB_KEY_D;
        case ' ': return KB_KEY_SPACE; case 'Z': return KB_KEY_LEFT_SHIFT;
        default: return KB_KEY_UNKNOWN;
    }
}

/* Called at the top of every loop iteration (rayinteractive.c:183): deliver this frame's scripted key on the
 * caller's thread, exactly where real minifb runs the keyboard callback. */
bool mfb_wait_sync(struct mfb_window* w) {
    if (!w) return false;
    if (w->frames_left <= 0) {
        double dt = w->t_last - w->t_first;
        if (w->frames_done > 1 && dt > 0)
            printf("minifb-stub: %ld frames, %.3f s, %.1f frames/s (launch + wait + read-back + present)\n", w->frames_done, dt,
                   (w->frames_done - 1) / dt);
        const char* dump = getenv("MFB_STUB_DUMP");
        if (dump && w->frames_done > 0) clw_host_write_png(dump, w->last, w->width, w->height);
        fflush(stdout);
        return false;
    }
    if (w->keys && w->keys[0] && w->on_key) {
        size_t n = strlen(w->keys);
        mfb_key k = decode(w->keys[w->key_pos++ % n]);
        if (k != KB_KEY_UNKNOWN) w->on_key(w, k, KB_MOD_NONE, true);
    }
    return true;
}

mfb_update_state mfb_update_ex(struct mfb_window* w, void* buffer, unsigned width, unsigned height) {
    if (!w) return STATE_INVALID_WINDOW;
    if (!buffer) return STATE_INVALID_BUFFER;
    if (width == w->width && height == w->height) memcpy(w->last, buffer, (size_t)width * height * 4);
    double t = now();
    if (w->frames_done == 0) w->t_first = t;
    w->t_last = t;
    w->frames_done++;
    w->frames_left--;
    return STATE_OK;
}

struct mfb_timer* mfb_timer_create(void) {
    struct mfb_timer* t = (struct mfb_timer*)malloc(sizeof *t);
    t->t0 = now();
    return t;
}
void mfb_timer_destroy(struct mfb_timer* tmr) { free(tmr); }

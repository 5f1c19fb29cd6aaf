/*
 * MiniFB.h -- headless stand-in for the minifb API surface the reference's interactive driver uses
 * (reference rayinteractive.c:4,32-33,118,122,181,183,191,199; keys :37-75).  dependencies/minifb is an empty,
 * un-vendored submodule in the reference, so rayinteractive.c cannot be built without it.
 * This stub opens no window: it replays a scripted key sequence and counts frames, which gives the
 * frames/s of the unchanged interactive loop on a GPU box.  Test / measurement infrastructure only.
 *
 *   MFB_STUB_FRAMES=<n>   frames before mfb_wait_sync returns false (default 100)
 *   MFB_STUB_KEYS=<str>   one key per frame, cycled: U D L R (arrows) W S A D' ' (space) Z (left shift), '.' = none
 *                         (note: 'D' is KB_KEY_D; arrows are u/d/l/r lower-case)
 *   MFB_STUB_DUMP=<path>  write the last presented frame as PNG (via clw_host_write_png)
 */
#ifndef MINIFB_STUB_H
#define MINIFB_STUB_H
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct mfb_window;
struct mfb_timer;

typedef enum {
    KB_KEY_UNKNOWN = -1,
    KB_KEY_SPACE = 32,
    KB_KEY_A = 65, KB_KEY_D = 68, KB_KEY_S = 83, KB_KEY_W = 87,
    KB_KEY_RIGHT = 262, KB_KEY_LEFT = 263, KB_KEY_DOWN = 264, KB_KEY_UP = 265,
    KB_KEY_LEFT_SHIFT = 340
} mfb_key;

typedef enum { KB_MOD_NONE = 0, KB_MOD_SHIFT = 1 } mfb_key_mod;

typedef enum { STATE_OK = 0, STATE_EXIT = -1, STATE_INVALID_WINDOW = -2, STATE_INVALID_BUFFER = -3, STATE_INTERNAL_ERROR = -4 } mfb_update_state;

typedef void (*mfb_keyboard_func)(struct mfb_window* window, mfb_key key, mfb_key_mod mod, bool isPressed);

struct mfb_window* mfb_open_ex(const char* title, unsigned width, unsigned height, unsigned flags);
void mfb_set_keyboard_callback(struct mfb_window* window, mfb_keyboard_func callback);
bool mfb_wait_sync(struct mfb_window* window);
mfb_update_state mfb_update_ex(struct mfb_window* window, void* buffer, unsigned width, unsigned height);
struct mfb_timer* mfb_timer_create(void);
void mfb_timer_destroy(struct mfb_timer* tmr);

#ifdef __cplusplus
}
#endif
#endif

/* Hand-written equivalent of the jconfig.h the reference's third_party/jpeg/CMakeLists.txt:10 generates from jconfig.h.cmake on this
 * platform (stddef.h and stdlib.h present; not Windows). TEST INFRASTRUCTURE ONLY (oracle/_ref/libref_features.so). */
#define HAVE_PROTOTYPES
#define HAVE_UNSIGNED_CHAR
#define HAVE_UNSIGNED_SHORT
#undef CHAR_IS_UNSIGNED
#define HAVE_STDDEF_H
#define HAVE_STDLIB_H
#undef NEED_BSD_STRINGS
#undef NEED_SYS_TYPES_H
#undef NEED_FAR_POINTERS
#undef NEED_SHORT_EXTERNAL_NAMES
#undef INCOMPLETE_TYPES_BROKEN
#ifdef JPEG_INTERNALS
#undef RIGHT_SHIFT_IS_UNSIGNED
#endif

/* Declarations for `cc -fsyntax-only -DLAMP_JNI_SYNTAX_CHECK jni/aten_jni.c` in images WITHOUT a JDK.
 * NOT a replacement for jni.h: only the members the generated shim uses, in no particular ABI order.  A real build includes the JDK's
 * <jni.h> (the #ifdef at the top of aten_jni.c). */
#ifndef LAMP_JNI_SYNTAX_CHECK_H
#define LAMP_JNI_SYNTAX_CHECK_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef int16_t jshort; typedef float jfloat;
typedef double jdouble; typedef jint jsize;
typedef struct jobject_* jobject;
typedef jobject jclass, jstring, jarray, jlongArray, jdoubleArray, jfloatArray, jintArray, jshortArray, jbyteArray, jbooleanArray, jthrowable;
#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2
#define JNI_TRUE 1
#define JNI_FALSE 0
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
  void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
  jlongArray (*NewLongArray)(JNIEnv*, jsize);
  jbyteArray (*NewByteArray)(JNIEnv*, jsize);
  void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
#define LAMP_JNI_ARR(T, N)                                             \
  T* (*Get##N##ArrayElements)(JNIEnv*, T##Array, jboolean*);           \
  void (*Release##N##ArrayElements)(JNIEnv*, T##Array, T*, jint);
  LAMP_JNI_ARR(jlong, Long) LAMP_JNI_ARR(jdouble, Double) LAMP_JNI_ARR(jfloat, Float) LAMP_JNI_ARR(jint, Int) LAMP_JNI_ARR(jshort, Short)
  LAMP_JNI_ARR(jbyte, Byte) LAMP_JNI_ARR(jboolean, Boolean)
#undef LAMP_JNI_ARR
};
#endif
